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
            raise KeyError(f"manifest {path}: no 'ID' column (every row needs a unique ID)")
        missing = [c for c in CSV_COLUMNS if c not in rd.fieldnames]
        if missing:
            raise KeyError(f"manifest {path} lacks the columns {missing}")
        for row in rd:
            if row["ID"] in seen:
                raise ValueError(f"manifest {path}: ID {row['ID']!r} appears twice")
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
    """Stack items that differ only in their leading extent: item ``i`` occupies ``out[i, :len_i]``, the rest of the row
    holds ``value``.  Returns ``(out, len_i / longest)``.  An empty list is an ``IndexError``; items of different rank or
    different trailing extents cannot share a row layout (``IndexError`` / ``EnvironmentError``, the exception types the
    reference's callers catch)."""
    count = len(tensors)
    if count == 0:
        raise IndexError("batch_pad_right: nothing to batch (empty list)")
    first = tensors[0]
    if count == 1:
        return first.unsqueeze(0), torch.ones(1)
    rank, tail = first.ndim, tuple(first.shape[1:])
    if all(t.ndim != rank for t in tensors[1:]):     # the reference lets a batch through as long as ONE other item agrees
        raise IndexError(f"batch_pad_right: no other item has the first item's rank ({rank})")
    for idx, t in enumerate(tensors):
        if t.ndim == rank and tuple(t.shape[1:]) != tail:
            raise EnvironmentError(f"batch_pad_right: item {idx} has trailing shape {tuple(t.shape[1:])}, the batch has {tail}")
    extents = [int(t.shape[0]) for t in tensors]
    longest = max(extents)
    if mode == "constant":
        out = first.new_full((count, longest) + tail, value)
        for row, t, n in zip(out, tensors, extents):
            row[:n] = t
    else:  # reflect / replicate / circular: torch's own padding, one item at a time
        grow = lambda t, n: torch.nn.functional.pad(t, [0, 0] * (rank - 1) + [0, longest - n], mode=mode)  # noqa: E731
        out = torch.stack([grow(t, n) for t, n in zip(tensors, extents)])
    return out, torch.tensor([n / longest for n in extents])


class PaddedData(tuple):
    """``(data, lengths)`` pair that also answers ``.data`` / ``.lengths``."""
    __slots__ = ()

    def __new__(cls, data, lengths):
        return super().__new__(cls, (data, lengths))

    data = property(lambda self: self[0])
    lengths = property(lambda self: self[1])

    def to(self, *args, **kwargs):
        return PaddedData(self[0].to(*args, **kwargs), self[1].to(*args, **kwargs))


def _as_tensor_if_array(v):
    """numeric numpy arrays become tensors (what torch's default_convert does); strings / objects are left alone"""
    kind = getattr(getattr(v, "dtype", None), "kind", None)
    return torch.as_tensor(v) if type(v).__module__ == "numpy" and kind is not None and kind not in "USO" else v


def _gather_plain(column):
    """a column that is not padded: Python numbers -> one tensor (the recipes read ``batch.cur_utter.item()``), same-shape
    tensors -> stacked, anything else (ids, paths, ragged tensors) stays a list"""
    head = column[0]
    if isinstance(head, (bool, int)):
        return torch.tensor(column)
    if isinstance(head, float):
        return torch.tensor(column, dtype=torch.float64)
    if isinstance(head, torch.Tensor) and all(isinstance(v, torch.Tensor) and v.shape == head.shape for v in column):
        return torch.stack(list(column), 0)
    return column


class PaddedBatch:
    """A batch as the recipes consume it (the contract of ``speechbrain/dataio/batch.py:101-178``): built from a list of
    example dicts, one COLUMN per key held in ``self._columns``.  Tensor-valued columns (or exactly those named in
    ``padded_keys``) become ``PaddedData(data, lengths)`` via ``batch_pad_right``; the others are gathered by
    ``_gather_plain``.  Columns read as attributes and items (``batch.sig`` -> ``(wavs, wav_lens)``, as
    ``compute_forward`` unpacks it), iterate in key order, and ``to(device)`` moves the tensor-valued ones (or those named
    in ``device_prep_keys``)."""

    def __init__(self, examples: Sequence[dict], padded_keys=None, device_prep_keys=None, padding_kwargs=None):
        columns, movable = {}, []
        for name in examples[0]:
            column = [_as_tensor_if_array(ex[name]) for ex in examples]
            tensor_valued = isinstance(column[0], torch.Tensor)
            pad = tensor_valued if padded_keys is None else name in padded_keys
            columns[name] = PaddedData(*batch_pad_right(column, **(padding_kwargs or {}))) if pad else _gather_plain(column)
            if tensor_valued if device_prep_keys is None else name in device_prep_keys:
                movable.append(name)
        object.__setattr__(self, "_columns", columns)
        object.__setattr__(self, "_movable", movable)
        object.__setattr__(self, "_size", len(examples))

    def __len__(self):
        return self._size

    def __getattr__(self, name):  # only reached for names that are not real attributes: the columns
        try:
            return self.__dict__["_columns"][name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        if name in self._columns:
            self._columns[name] = value
        else:
            object.__setattr__(self, name, value)

    def __getitem__(self, name):
        if name not in self._columns:
            raise KeyError(f"this batch has no column {name!r} (columns: {list(self._columns)})")
        return self._columns[name]

    def __iter__(self):
        return iter(self._columns.values())

    def to(self, *args, **kwargs):
        for name in self._movable:
            held = self._columns[name]
            if isinstance(held, (PaddedData, torch.Tensor)):
                self._columns[name] = held.to(*args, **kwargs)
        return self
