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
    """Ranking function of the default selection: the save time recorded in the meta file."""
    return ckpt.meta["unixtime"]


def torch_recovery(obj, path, end_of_epoch=True, device=None):
    """Default loader of a ``<name>.ckpt`` file: a torch-saved state dict, loaded strictly where the object takes ``strict``."""
    del end_of_epoch
    state = torch.load(path, map_location=device or "cpu")
    try:
        obj.load_state_dict(state, strict=True)
    except TypeError:
        obj.load_state_dict(state)


def _read_checkpoint_dir(folder: pathlib.Path) -> Optional[Checkpoint]:
    """``CKPT+<name>/`` with its ``CKPT.yaml`` -> record; anything else in the save folder -> None."""
    meta_file = folder / METAFNAME
    if not (folder.is_dir() and folder.name.startswith(CKPT_PREFIX) and meta_file.exists()):
        return None
    with open(meta_file) as fh:
        meta = yaml.safe_load(fh)
    return Checkpoint(folder, meta, {f.stem: f for f in folder.glob("*" + PARAMFILE_EXT)})


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
        if not isinstance(recoverables, collections.abc.Mapping):
            raise AttributeError(f"recoverables must map names to objects, got {type(recoverables).__name__}")
        self.recoverables.update(recoverables)

    # ---- listing / selection ----
    def list_checkpoints(self) -> List[Checkpoint]:
        if not self.checkpoints_dir.is_dir():
            return []
        found = (_read_checkpoint_dir(d) for d in self.checkpoints_dir.iterdir())
        return [c for c in found if c is not None]

    def find_checkpoints(self, importance_key=None, max_key=None, min_key=None, ckpt_predicate=None,
                         max_num_checkpoints=None) -> List[Checkpoint]:
        """Checkpoints best-first.  Ranking: ``importance_key(ckpt)`` descending, or the meta entry ``max_key`` descending
        / ``min_key`` ascending (then only checkpoints whose meta carries that entry take part; ``max_key`` wins if both
        names are given, as in the reference), or — with nothing given — save time.  Equal ranks: latest first.  A
        ranking function together with a key name is the one combination refused."""
        named = max_key or min_key
        if importance_key is not None and named:
            raise ValueError("pass either importance_key or one of max_key / min_key, not both")
        if importance_key is not None:
            score, needs = importance_key, None
        elif max_key:
            score, needs = (lambda c: c.meta[max_key]), max_key
        elif min_key:
            score, needs = (lambda c: -c.meta[min_key]), min_key
        else:
            score, needs = ckpt_recency, None
        pool = [c for c in self.list_checkpoints()
                if (needs is None or needs in c.meta) and (ckpt_predicate is None or ckpt_predicate(c))]
        pool.sort(key=ckpt_recency, reverse=True)   # list.sort is stable: the second sort keeps latest-first among equals
        pool.sort(key=score, reverse=True)
        return pool if max_num_checkpoints is None else pool[:max_num_checkpoints]

    def find_checkpoint(self, importance_key=None, max_key=None, min_key=None, ckpt_predicate=None) -> Optional[Checkpoint]:
        ranked = self.find_checkpoints(importance_key, max_key, min_key, ckpt_predicate, max_num_checkpoints=1)
        return ranked[0] if ranked else None

    # ---- loading ----
    def load_checkpoint(self, checkpoint: Checkpoint, device=None):
        """Every registered object takes its ``<name>.ckpt``: through its custom hook if one was registered, else through
        ``torch_recovery`` if it has ``load_state_dict``.  A registered name without a file is an error unless
        ``allow_partial_load``."""
        at_epoch_end = checkpoint.meta["end-of-epoch"]
        for name, obj in self.recoverables.items():
            source = checkpoint.paramfiles.get(name)
            if source is None:
                if not self.allow_partial_load:
                    raise RuntimeError(f"{checkpoint.path} holds no {name}{PARAMFILE_EXT} for the registered object {name!r}")
                continue
            loader = self.custom_load_hooks.get(name)
            if loader is None:
                if not hasattr(obj, "load_state_dict"):
                    raise RuntimeError(f"{name!r} ({type(obj).__name__}) has neither a custom load hook nor load_state_dict")
                loader = torch_recovery
            loader(obj, source, at_epoch_end, device)

    def recover_if_possible(self, importance_key=None, max_key=None, min_key=None, ckpt_predicate=None, device=None):
        chosen = self.find_checkpoint(importance_key, max_key, min_key, ckpt_predicate)
        if chosen is not None:
            self.load_checkpoint(chosen, device)
        return chosen
