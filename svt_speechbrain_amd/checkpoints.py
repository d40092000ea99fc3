"""Reader for SpeechBrain checkpoint directories — loads the authors' released ``save/CKPT+*/{wav2vec2,model}.ckpt``
files unchanged into this package's modules.

Same selection semantics as ``speechbrain.utils.checkpoints.Checkpointer`` (``find_checkpoints`` :697-790,
``recover_if_possible`` :792-846, ``_call_load_hooks`` :963-1003, ``torch_recovery`` :69-95): a checkpoint is a
directory ``CKPT+<name>`` holding ``CKPT.yaml`` (meta: ``unixtime``, ``end-of-epoch`` and whatever the recipe added,
e.g. ``loss`` / ``COnPOff_f1`` — ``MIR_ST500/train_audio_ssl.py:178-186``) and one ``<recoverable>.ckpt`` per registered
object; with no key the most recent is chosen, ``max_key`` / ``min_key`` consider only checkpoints that carry the key,
ties go to the most recent.  Read-only: saving stays with the training framework.
"""
from __future__ import annotations

import collections
import pathlib
from typing import Callable, Dict, List, Mapping, Optional

import torch
import yaml

CKPT_PREFIX = "CKPT"
METAFNAME = f"{CKPT_PREFIX}.yaml"
PARAMFILE_EXT = ".ckpt"

Checkpoint = collections.namedtuple("Checkpoint", ["path", "meta", "paramfiles"])
Checkpoint.__hash__ = lambda self: hash(self.path)


def ckpt_recency(ckpt: Checkpoint):
    """Importance key: later ``unixtime`` is more important."""
    return ckpt.meta["unixtime"]


def torch_recovery(obj, path, end_of_epoch=True, device=None):
    """``obj.load_state_dict(torch.load(path))``, strict when the object's load_state_dict accepts it."""
    del end_of_epoch
    state = torch.load(path, map_location=device or "cpu")
    try:
        obj.load_state_dict(state, strict=True)
    except TypeError:
        obj.load_state_dict(state)


def _is_checkpoint_dir(path: pathlib.Path) -> bool:
    path = pathlib.Path(path)
    return path.is_dir() and path.name.startswith(CKPT_PREFIX) and (path / METAFNAME).exists()


class Checkpointer:
    """Read-side of the SpeechBrain ``Checkpointer``: ``Checkpointer(save_folder, {"wav2vec2": enc, "model": head})``
    then ``recover_if_possible(min_key="loss")`` — the call ``Brain.evaluate`` makes (``speechbrain/core.py:1279-1283``)."""

    def __init__(self, checkpoints_dir, recoverables: Optional[Mapping] = None, custom_load_hooks: Optional[Mapping] = None,
                 allow_partial_load: bool = False):
        self.checkpoints_dir = pathlib.Path(checkpoints_dir)
        self.recoverables: Dict[str, object] = {}
        if recoverables is not None:
            self.add_recoverables(recoverables)
        self.custom_load_hooks: Dict[str, Callable] = dict(custom_load_hooks or {})
        self.allow_partial_load = allow_partial_load

    def add_recoverable(self, name, obj, custom_load_hook=None):
        self.recoverables[name] = obj
        if custom_load_hook is not None:
            self.custom_load_hooks[name] = custom_load_hook

    def add_recoverables(self, recoverables):
        if isinstance(recoverables, collections.abc.Mapping):
            self.recoverables.update(recoverables)
        else:
            raise AttributeError(f"Checkpointer needs a mapping (e.g. dict), got {recoverables!r} instead.")

    # ---- listing / selection ----
    def list_checkpoints(self) -> List[Checkpoint]:
        out = []
        if not self.checkpoints_dir.is_dir():
            return out
        for d in self.checkpoints_dir.iterdir():
            if not _is_checkpoint_dir(d):
                continue
            with open(d / METAFNAME) as fi:
                meta = yaml.load(fi, Loader=yaml.SafeLoader)
            files = {f.stem: f for f in d.iterdir() if f.suffix == PARAMFILE_EXT}
            out.append(Checkpoint(d, meta, files))
        return out

    def find_checkpoints(self, importance_key=None, max_key=None, min_key=None, ckpt_predicate=None,
                         max_num_checkpoints=None) -> List[Checkpoint]:
        if importance_key is None and min_key is None and max_key is None:
            importance_key = ckpt_recency
        if max_key and not importance_key:
            def importance_key(ckpt):
                return ckpt.meta[max_key]
            user_pred = ckpt_predicate

            def ckpt_predicate(ckpt):
                return max_key in ckpt.meta and (user_pred is None or user_pred(ckpt))
        elif min_key and not importance_key:
            def importance_key(ckpt):
                return -ckpt.meta[min_key]
            user_pred = ckpt_predicate

            def ckpt_predicate(ckpt):
                return min_key in ckpt.meta and (user_pred is None or user_pred(ckpt))
        elif min_key or max_key:
            raise ValueError("Must specify only one of 'importance_key', 'max_key', and 'min_key'.")
        ckpts = list(filter(ckpt_predicate, self.list_checkpoints()))
        ckpts = sorted(ckpts, key=ckpt_recency, reverse=True)  # stable: importance ties go to the most recent
        ranked = sorted(ckpts, key=importance_key, reverse=True)
        return ranked if max_num_checkpoints is None else ranked[:max_num_checkpoints]

    def find_checkpoint(self, importance_key=None, max_key=None, min_key=None, ckpt_predicate=None) -> Optional[Checkpoint]:
        found = self.find_checkpoints(importance_key, max_key, min_key, ckpt_predicate)
        return found[0] if found else None

    # ---- loading ----
    def load_checkpoint(self, checkpoint: Checkpoint, device=None):
        end_of_epoch = checkpoint.meta["end-of-epoch"]
        for name, obj in self.recoverables.items():
            try:
                loadpath = checkpoint.paramfiles[name]
            except KeyError:
                if self.allow_partial_load:
                    continue
                raise RuntimeError(f"Loading checkpoint from {checkpoint.path}, but missing a load path for {name}")
            if name in self.custom_load_hooks:
                self.custom_load_hooks[name](obj, loadpath, end_of_epoch, device)
            elif hasattr(obj, "load_state_dict"):
                torch_recovery(obj, loadpath, end_of_epoch, device)
            else:
                raise RuntimeError(f"Don't know how to load {type(obj)}. Register default hook or add custom hook for this object.")

    def recover_if_possible(self, importance_key=None, max_key=None, min_key=None, ckpt_predicate=None, device=None):
        chosen = self.find_checkpoint(importance_key, max_key, min_key, ckpt_predicate)
        if chosen is not None:
            self.load_checkpoint(chosen, device)
        return chosen
