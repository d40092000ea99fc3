"""Input side of the recipes: the CSV manifest, the 5 s utterance slicing rule and the
right-zero-padded batch with relative lengths.  Host code, as in the reference.

* manifest columns ``ID,duration,wav,utter_id,utter_num,frame_anno,song_anno`` written by
  ``MIR_ST500/prepare_benchmarks.py:104-135`` (``prepare_csv_benchmarks``); ``plan_utterances`` restates its
  ``utter_num = round(duration / dur_thrd)`` rule (Python banker's rounding, last utterance takes the remainder).
* ``slice_audio`` / ``slice_annotation`` are the two dynamic-item pipelines of ``MIR_ST500/train_audio_ssl.py:381-421``
  (``round()`` of ``(utter_id-1) * rate * dur_threshold``; the last utterance runs to the end).
* ``batch_pad_right`` / ``PaddedBatch`` follow ``speechbrain/utils/data_utils.py:361-424`` and
  ``speechbrain/dataio/batch.py:101-137``: pad dim 0 to the longest item, relative length = len / max_len.
"""
from __future__ import annotations

import csv
from typing import Dict, List, Sequence

import torch

CSV_COLUMNS = ["ID", "duration", "wav", "utter_id", "utter_num", "frame_anno", "song_anno"]


def read_manifest(path: str) -> List[Dict[str, str]]:
    """Rows of a recipe manifest, keyed like ``DynamicItemDataset.from_csv`` (``ID`` is mandatory and unique;
    ``duration`` becomes a float, everything else stays a string)."""
    rows = []
    seen = set()
    with open(path, newline="") as fh:
        rd = csv.DictReader(fh, skipinitialspace=True)
        if rd.fieldnames is None or "ID" not in rd.fieldnames:
            raise KeyError("CSV has to have an 'ID' field, with unique ids for all data points")
        missing = [c for c in CSV_COLUMNS if c not in rd.fieldnames]
        if missing:
            raise KeyError(f"manifest {path} lacks the columns {missing}")
        for row in rd:
            if row["ID"] in seen:
                raise ValueError(f"Duplicate id: {row['ID']}")
            seen.add(row["ID"])
            row = dict(row)
            row["duration"] = float(row["duration"])
            rows.append(row)
    return rows


def plan_utterances(duration: float, dur_thrd: float = 5) -> List[float]:
    """Durations of the utterances a song of ``duration`` seconds is cut into (prepare_benchmarks.py:119-127)."""
    utter_num = round(duration / dur_thrd)
    out = []
    for i in range(1, utter_num + 1):
        if i == utter_num:
            dur = duration - (utter_num - 1) * dur_thrd
            assert 0 < dur <= dur_thrd * 3 / 2
        else:
            dur = dur_thrd
        out.append(dur)
    return out


def _bounds(utter_id: int, utter_num: int, rate: float, dur_threshold: float):
    if utter_id == utter_num:
        return round((utter_id - 1) * rate * dur_threshold), None
    return round((utter_id - 1) * rate * dur_threshold), round(utter_id * rate * dur_threshold)


def slice_audio(sig: torch.Tensor, utter_id, utter_num, sample_rate: int = 16000, dur_threshold: float = 5) -> torch.Tensor:
    """``audio_pipeline`` (train_audio_ssl.py:381-395): 1-D signal of the whole song -> this utterance's samples."""
    assert len(sig.shape) == 1
    lo, hi = _bounds(int(utter_id), int(utter_num), sample_rate, dur_threshold)
    return sig[lo:] if hi is None else sig[lo:hi]


def slice_annotation(anno: torch.Tensor, utter_id, utter_num, frame_rate: float = 49.8, dur_threshold: float = 5) -> torch.Tensor:
    """``anno_pipeline`` (train_audio_ssl.py:411-421): (frames, 4) song annotation -> this utterance's frames."""
    lo, hi = _bounds(int(utter_id), int(utter_num), frame_rate, dur_threshold)
    return anno[lo:] if hi is None else anno[lo:hi]


def batch_pad_right(tensors: Sequence[torch.Tensor], mode: str = "constant", value=0):
    """Right-pad dim 0 to the longest item and stack; second result = relative lengths (len / max_len)."""
    if not len(tensors):
        raise IndexError("Tensors list must not be empty")
    if len(tensors) == 1:
        return tensors[0].unsqueeze(0), torch.tensor([1.0])
    if not any(tensors[i].ndim == tensors[0].ndim for i in range(1, len(tensors))):
        raise IndexError("All tensors must have same number of dimensions")
    for dim in range(1, tensors[0].ndim):
        if not all(x.shape[dim] == tensors[0].shape[dim] for x in tensors[1:]):
            raise EnvironmentError("Tensors should have same dimensions except for the first one")
    max_len = max(x.shape[0] for x in tensors)
    batched, valid = [], []
    for t in tensors:
        pads = [0, 0] * (t.ndim - 1) + [0, max_len - t.shape[0]]
        batched.append(torch.nn.functional.pad(t, pads, mode=mode, value=value))
        valid.append(t.shape[0] / max_len)
    return torch.stack(batched), torch.tensor(valid)


class PaddedData(tuple):
    """``(data, lengths)`` pair, attribute access like speechbrain's namedtuple."""
    __slots__ = ()

    def __new__(cls, data, lengths):
        return super().__new__(cls, (data, lengths))

    data = property(lambda self: self[0])
    lengths = property(lambda self: self[1])


class PaddedBatch:
    """Collate a list of example dicts: tensor-valued keys are right-padded into ``PaddedData(data, lengths)``, other
    keys become lists (``speechbrain/dataio/batch.py:101-137``).  ``batch.sig`` -> ``(wavs, wav_lens)`` as
    ``compute_forward`` unpacks it; ``to(device)`` moves the padded tensors."""

    def __init__(self, examples: Sequence[dict], padded_keys=None, device_prep_keys=None, padding_kwargs=None):
        self.__length = len(examples)
        self.__keys = list(examples[0].keys())
        self.__padded_keys = []
        self.__device_prep_keys = []
        for key in self.__keys:
            values = [ex[key] for ex in examples]
            # default_convert: numpy arrays become tensors, everything else is left alone
            values = [torch.as_tensor(v) if type(v).__module__ == "numpy" and getattr(v, "dtype", None) is not None
                      and v.dtype.kind not in "USO" else v for v in values]
            if (padded_keys is not None and key in padded_keys) or (padded_keys is None and isinstance(values[0], torch.Tensor)):
                self.__padded_keys.append(key)
                setattr(self, key, PaddedData(*batch_pad_right(values, **(padding_kwargs or {}))))
            else:
                # mod_default_collate: Python numbers become one tensor (batch.cur_utter.item() in the recipes), equal-size
                # tensors are stacked, strings and ragged items stay lists
                if isinstance(values[0], bool) or isinstance(values[0], (int, float)):
                    values = torch.tensor(values, dtype=torch.float64 if isinstance(values[0], float) else None)
                elif isinstance(values[0], torch.Tensor):
                    try:
                        values = torch.stack(values, 0)
                    except RuntimeError:
                        pass
                setattr(self, key, values)
            if (device_prep_keys is not None and key in device_prep_keys) or (device_prep_keys is None and isinstance(values[0], torch.Tensor)):
                self.__device_prep_keys.append(key)

    def __len__(self):
        return self.__length

    def __getitem__(self, key):
        if key in self.__keys:
            return getattr(self, key)
        raise KeyError(f"Batch doesn't have key: {key}")

    def __iter__(self):
        return iter(getattr(self, key) for key in self.__keys)

    def to(self, *args, **kwargs):
        for key in self.__device_prep_keys:
            v = getattr(self, key)
            if isinstance(v, PaddedData):
                setattr(self, key, PaddedData(v.data.to(*args, **kwargs), v.lengths.to(*args, **kwargs)))
            elif isinstance(v, torch.Tensor):
                setattr(self, key, v.to(*args, **kwargs))
        return self
