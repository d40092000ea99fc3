"""Note-level scoring of a transcription — the numbers the recipes read from ``mir_eval.transcription.evaluate``:
COnPOff (onset + pitch + offset), COnP (``*_no_offset``) and COn (``Onset_*``) precision / recall / F-measure
(``MIR_ST500/train_audio_ssl.py:119-134``), plus COff (``Offset_*``, offsets alone), which the N20EMv2 recipes add
(``N20EMv2/audio_only/train_audio_ssl.py:148-150``), and mir_eval's two ``Average_Overlap_Ratio`` entries.

PARITY UNPINNED: ``mir_eval`` is a third-party dependency of the reference (``MIR_ST500/train_audio_ssl.py:14-15``, no
version pin) that is not installed in the build container, so nothing here could be checked against it.  The functions
restate mir_eval's published algorithm (``mir_eval/transcription.py`` ``match_notes`` / ``match_note_onsets`` /
``precision_recall_f1_overlap`` / ``onset_precision_recall_f1``, ``mir_eval/util.py`` ``_bipartite_match`` = Hopcroft-Karp):
``match_note_offsets`` / ``offset_precision_recall_f1`` / ``average_overlap_ratio``): distances rounded to 4 decimals,
hit when ``<=`` the tolerance, offsets within ``max(offset_ratio * ref_duration, offset_min_tolerance)``, pitch within
``pitch_tolerance`` cents, then a maximum bipartite matching.  Tests hold hand-derived known answers only.  The
precision / recall / F numbers depend on the SIZE of the maximum matching only (unique); the overlap ratios average over
the matched PAIRS, and two maximum matchings can pair notes differently, so those two may differ from mir_eval's in
ambiguous cases even once it is available.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

N_DECIMALS = 4


def midi_to_hz(midi):
    return 440.0 * (2.0 ** ((np.asarray(midi, dtype=float) - 69.0) / 12.0))


def _max_bipartite_matching(adj: Dict[int, List[int]]) -> Dict[int, int]:
    """Maximum matching of the bipartite graph est -> [ref...]; returns {ref: est}.  Augmenting paths (Kuhn): the
    matching SIZE is what the metrics use and it equals Hopcroft-Karp's."""
    match_ref: Dict[int, int] = {}

    def try_assign(e, seen):
        for r in adj.get(e, []):
            if r in seen:
                continue
            seen.add(r)
            if r not in match_ref or try_assign(match_ref[r], seen):
                match_ref[r] = e
                return True
        return False

    for e in sorted(adj):
        try_assign(e, set())
    return match_ref


def _validate(intervals, pitches=None):
    intervals = np.asarray(intervals, dtype=float).reshape(-1, 2)
    if intervals.size and (intervals[:, 1] - intervals[:, 0] <= 0).any():
        raise ValueError("All interval durations must be strictly positive")
    if intervals.size and (intervals < 0).any():
        raise ValueError("Negative interval times found")
    if pitches is not None:
        pitches = np.asarray(pitches, dtype=float).reshape(-1)
        if pitches.shape[0] != intervals.shape[0]:
            raise ValueError("Intervals and pitches have different lengths")
        if pitches.size and (pitches <= 0).any():
            raise ValueError("Pitches must be positive (Hz)")
    return intervals, pitches


def match_note_onsets(ref_intervals, est_intervals, onset_tolerance=0.05):
    d = np.around(np.abs(np.subtract.outer(ref_intervals[:, 0], est_intervals[:, 0])), N_DECIMALS)
    hits = np.where(d <= onset_tolerance)
    adj: Dict[int, List[int]] = {}
    for r, e in zip(*hits):
        adj.setdefault(int(e), []).append(int(r))
    return sorted(_max_bipartite_matching(adj).items())


def match_note_offsets(ref_intervals, est_intervals, offset_ratio=0.2, offset_min_tolerance=0.05):
    """mir_eval ``match_note_offsets``: offsets alone, tolerance ``max(offset_ratio * ref_duration, offset_min_tolerance)``."""
    d = np.around(np.abs(np.subtract.outer(ref_intervals[:, 1], est_intervals[:, 1])), N_DECIMALS)
    tol = np.maximum(offset_ratio * (ref_intervals[:, 1] - ref_intervals[:, 0]), offset_min_tolerance)
    adj: Dict[int, List[int]] = {}
    for r, e in zip(*np.where(d <= tol[:, None])):
        adj.setdefault(int(e), []).append(int(r))
    return sorted(_max_bipartite_matching(adj).items())


def average_overlap_ratio(ref_intervals, est_intervals, matching) -> float:
    """mir_eval ``average_overlap_ratio``: mean over matched (ref, est) pairs of intersection / union of the two intervals."""
    ratios = []
    for r, e in matching:
        ri, ei = ref_intervals[r], est_intervals[e]
        ratios.append((min(ri[1], ei[1]) - max(ri[0], ei[0])) / (max(ri[1], ei[1]) - min(ri[0], ei[0])))
    return float(np.mean(ratios)) if ratios else 0.0


def match_notes(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance=0.05, pitch_tolerance=50.0,
                offset_ratio=0.2, offset_min_tolerance=0.05):
    onset_hit = np.around(np.abs(np.subtract.outer(ref_intervals[:, 0], est_intervals[:, 0])), N_DECIMALS) <= onset_tolerance
    pitch_hit = np.abs(1200 * np.subtract.outer(np.log2(ref_pitches), np.log2(est_pitches))) <= pitch_tolerance
    hit = onset_hit & pitch_hit
    if offset_ratio is not None:
        off_d = np.around(np.abs(np.subtract.outer(ref_intervals[:, 1], est_intervals[:, 1])), N_DECIMALS)
        ref_dur = ref_intervals[:, 1] - ref_intervals[:, 0]
        tol = np.maximum(offset_ratio * ref_dur, offset_min_tolerance)
        hit &= off_d <= tol[:, None]
    adj: Dict[int, List[int]] = {}
    for r, e in zip(*np.where(hit)):
        adj.setdefault(int(e), []).append(int(r))
    return sorted(_max_bipartite_matching(adj).items())


def _prf(n_match, n_ref, n_est):
    if n_ref == 0 or n_est == 0:
        return 0.0, 0.0, 0.0
    p, r = n_match / n_est, n_match / n_ref
    f = 0.0 if p + r == 0 else 2 * p * r / (p + r)
    return p, r, f


def evaluate(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance=0.05, pitch_tolerance=50.0,
             offset_ratio=0.2, offset_min_tolerance=0.05) -> Dict[str, float]:
    """Every key of ``mir_eval.transcription.evaluate`` (``Precision`` ... ``Offset_F-measure``); pitches in Hz, intervals
    in seconds."""
    ref_intervals, ref_pitches = _validate(ref_intervals, ref_pitches)
    est_intervals, est_pitches = _validate(est_intervals, est_pitches)
    n_ref, n_est = len(ref_intervals), len(est_intervals)
    out = {}
    if n_ref == 0 or n_est == 0:
        m_full = m_nooff = m_on = m_off = []
    else:
        m_full = match_notes(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance, pitch_tolerance,
                             offset_ratio, offset_min_tolerance)
        m_nooff = match_notes(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance, pitch_tolerance, None)
        m_on = match_note_onsets(ref_intervals, est_intervals, onset_tolerance)
        m_off = match_note_offsets(ref_intervals, est_intervals, offset_ratio, offset_min_tolerance)
    out["Precision"], out["Recall"], out["F-measure"] = _prf(len(m_full), n_ref, n_est)
    out["Average_Overlap_Ratio"] = average_overlap_ratio(ref_intervals, est_intervals, m_full)
    out["Precision_no_offset"], out["Recall_no_offset"], out["F-measure_no_offset"] = _prf(len(m_nooff), n_ref, n_est)
    out["Average_Overlap_Ratio_no_offset"] = average_overlap_ratio(ref_intervals, est_intervals, m_nooff)
    out["Onset_Precision"], out["Onset_Recall"], out["Onset_F-measure"] = _prf(len(m_on), n_ref, n_est)
    out["Offset_Precision"], out["Offset_Recall"], out["Offset_F-measure"] = _prf(len(m_off), n_ref, n_est)
    return out


def score_song(est_notes, ref_notes, onset_tolerance=0.05, pitch_tolerance=50.0) -> Dict[str, float]:
    """``[[onset_s, offset_s, midi], ...]`` lists (``frame2note`` output / ``annotation.json``) -> the metrics, with
    the recipe's ``midi_to_hz`` conversion (train_audio_ssl.py:112-117)."""
    est = np.asarray(est_notes, dtype=float).reshape(-1, 3)
    ref = np.asarray(ref_notes, dtype=float).reshape(-1, 3)
    return evaluate(ref[:, :2], midi_to_hz(ref[:, 2]), est[:, :2], midi_to_hz(est[:, 2]), onset_tolerance, pitch_tolerance)
