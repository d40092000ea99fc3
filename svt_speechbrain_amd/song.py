"""Song-level driver either side of the hot path.

* utterance slicing of a song — the rule of ``MIR_ST500/prepare_benchmarks.py:117-126`` (``utter_num =
  round(duration / 5 s)``, last utterance takes the remainder, up to 7.5 s) and of the recipe's audio pipeline
  (``MIR_ST500/train_audio_ssl.py:373-390``: sample bounds ``round((i-1) * sr * dur)`` .. ``round(i * sr * dur)``);
* per-utterance forward with batch = 1 (the reference asserts it, ``train_audio_ssl.py:90``), frame predictions
  concatenated over the song, ``frame2note`` once at the last utterance (``:85-107``);
* the stage-1 feature writer of ``N20EMv2/audio_only/extract_ssl_feats.py:102-116``: per-utterance ``feats[0]``
  concatenated along time and ``torch.save``d as ``<folder>/noise_data/clean_feats.pt`` (or
  ``noise_data/<type>/SNR_<db>dB_feats.pt``) — the input of the audio-visual recipe;
* its video twin, ``N20EMv2/video_only/extract_ssl_feats.py:28-36,99-111``: the song's lip ROI (``np.load``: ``(T, H, W)`` uint8),
  ``transform_eval``, utterances of ``dur_threshold`` seconds at the video's ``sample_rate`` (50 frames/s; bounds as for the audio,
  ``train_video_ssl.py:537-546``), batch-1 AV-HuBERT forwards, ``feats[0]`` concatenated and saved as
  ``<folder>/noise_data/video_feats.pt`` (``song_video_features`` / ``save_song_video_features``).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch

from .decode import decode_frames, frame2note, frames2note, frames2note_batch, frames_to_info


def utterance_bounds(n_samples: int, sample_rate: int = 16000, dur_threshold: float = 5.0) -> List[Tuple[int, int]]:
    """Sample ranges of the utterances of one song (1-based ids ``1..utter_num`` in the reference's CSV)."""
    duration = n_samples / sample_rate
    utter_num = round(duration / dur_threshold)
    if utter_num < 1:
        utter_num = 1
    out = []
    for i in range(1, utter_num + 1):
        lo = round((i - 1) * sample_rate * dur_threshold)
        hi = n_samples if i == utter_num else round(i * sample_rate * dur_threshold)
        out.append((lo, hi))
    return out


class SongTranscriber:
    """waveform of a whole song -> (notes, per-song features), utterance by utterance like the reference's eval loop."""

    def __init__(self, encoder, head, pitch_octave_num: int = 4, pitch_class_num: int = 12, onset_threshold: float = 0.4,
                 offset_threshold: float = 0.5, frame_rate: float = 49.8, sample_rate: int = 16000,
                 dur_threshold: float = 5.0, streams: int = 2, batch_utterances: bool = True, max_batch: int = 64):
        self.encoder, self.head = encoder, head
        self.pitch_octave_num, self.pitch_class_num = pitch_octave_num, pitch_class_num
        self.onset_threshold, self.offset_threshold = onset_threshold, offset_threshold
        self.frame_rate, self.sample_rate, self.dur_threshold = frame_rate, sample_rate, dur_threshold
        # Utterances are forwarded one at a time (batch 1, as the reference's eval asserts: the whole-batch norms would
        # couple the clips of a larger batch), but they are independent of each other: successive utterances go round-robin
        # to `streams` HIP streams, each with its own copy of the encoder object (device handle + workspace), and the frames
        # of the whole song are decoded by ONE kernel + ONE device-to-host copy at the end.
        self.streams = max(1, int(streams))
        # batch_utterances: the equal-length utterances of a song (all but the last are dur_threshold seconds long) go through
        # the encoder as ONE batch whose two whole-tensor norms are taken per clip (``clips_per_norm_group=1``): the same
        # numbers as the batch-1 forwards of the reference's evaluation loop, at the throughput of a batch
        self.batch_utterances = bool(batch_utterances) and hasattr(encoder, "replica")
        self.max_batch = max(1, int(max_batch))
        self._encoders = None
        self._side = None

    def _lanes(self, device):
        if self._encoders is None:
            if self.streams > 1 and not hasattr(self.encoder, "replica"):
                self.streams = 1  # a foreign encoder module: no second device object to run concurrently
            self._encoders = [self.encoder] + [self.encoder.replica() for _ in range(self.streams - 1)]
            self._side = [torch.cuda.Stream(device=device) for _ in range(self.streams - 1)]
        return self._encoders, [torch.cuda.current_stream(device)] + self._side

    @torch.no_grad()
    def transcribe(self, song: torch.Tensor, return_feats: bool = False):
        """song: 1-D waveform on the GPU.  Returns notes [[on_s, off_s, midi], ...] (and the (T_song, D) features)."""
        if song.dim() != 1:
            raise ValueError("expected a mono 1-D waveform")
        encs, lanes = self._lanes(song.device)
        main = lanes[0]
        for st in lanes[1:]:
            st.wait_stream(main)
        logits_all, feats_all = [], []
        bounds = utterance_bounds(song.shape[0], self.sample_rate, self.dur_threshold)
        # runs of consecutive utterances of one length (a multiple of 4 samples) -> one batched forward each
        jobs = []
        i = 0
        while i < len(bounds):
            n = bounds[i][1] - bounds[i][0]
            j = i + 1
            if self.batch_utterances and n % 4 == 0:
                while j < len(bounds) and j - i < self.max_batch and bounds[j][1] - bounds[j][0] == n and bounds[j][0] == bounds[j - 1][1]:
                    j += 1
            jobs.append((i, j))
            i = j
        for q, (i, j) in enumerate(jobs):
            k = q % len(lanes)
            lo, hi = bounds[i][0], bounds[j - 1][1]
            with torch.cuda.stream(lanes[k]):
                if j - i > 1:
                    feats = encs[k](song[lo:hi].view(j - i, -1), clips_per_norm_group=1)
                else:
                    feats = encs[k](song[lo:hi].unsqueeze(0))  # batch of one utterance, as in the reference eval
                logits = self.head(feats)
                if k:  # produced on a side stream, consumed (cat + decode) on the main one: tell the caching allocator
                    logits.record_stream(main)
                    feats.record_stream(main)
                logits_all.append(logits.reshape(-1, logits.shape[-1]))
                if return_feats:
                    feats_all.append(feats.reshape(-1, feats.shape[-1]))
        for st in lanes[1:]:
            main.wait_stream(st)
        frames = decode_frames(torch.cat(logits_all, dim=0), self.pitch_octave_num, self.pitch_class_num)
        notes = frames2note_batch(frames, self.onset_threshold, self.offset_threshold, 1 / self.frame_rate,
                                  pitch_octave_num=self.pitch_octave_num, pitch_class_num=self.pitch_class_num)[0]
        if return_feats:
            return notes, torch.cat(feats_all, dim=0)
        return notes


def feature_path(song_folder: str, add_noise: bool = False, noise_type: Optional[str] = None,
                 snr_db: Optional[int] = None) -> str:
    """Where the reference's extract pass stores the per-song features (extract_ssl_feats.py:108-115)."""
    if add_noise:
        return os.path.join(song_folder, "noise_data", str(noise_type), f"SNR_{snr_db}dB_feats.pt")
    return os.path.join(song_folder, "noise_data", "clean_feats.pt")


def save_song_features(feats: torch.Tensor, song_folder: str, add_noise: bool = False, noise_type: Optional[str] = None,
                       snr_db: Optional[int] = None) -> str:
    path = feature_path(song_folder, add_noise, noise_type, snr_db)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(feats.detach().cpu(), path)
    return path


def video_feature_path(song_folder: str) -> str:
    """``os.path.join(dirname(video_path), "noise_data", "video_feats.pt")`` (N20EMv2/video_only/extract_ssl_feats.py:104-110)."""
    return os.path.join(song_folder, "noise_data", "video_feats.pt")


def save_song_video_features(feats: torch.Tensor, song_folder: str) -> str:
    path = video_feature_path(song_folder)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(feats.detach().cpu(), path)
    return path


@torch.no_grad()
def song_video_features(encoder, roi: torch.Tensor, frame_rate: int = 50, dur_threshold: float = 5.0, transform=None) -> torch.Tensor:
    """Per-song video features as the reference's extraction pass computes them (extract_ssl_feats.py:28-36, 99-107): ``roi`` is the
    song's ``(T, H, W)`` **uint8** lip ROI on the GPU (what ``np.load(video)`` returns); utterance ``i`` of ``round(T / (rate * dur))``
    covers frames ``round((i-1) * rate * dur) : round(i * rate * dur)`` (the last one runs to the end); each goes through ``encoder``
    (``FairseqAVHubertPretrain``) as a batch of ONE -- the wrapper's whole-tensor output norm is per utterance, as in the reference's
    batch-1 evaluation -- with ``transform_eval`` applied inside the front-end's padding kernel; returns the ``(T, D)`` concatenation."""
    if roi.dim() != 3 or roi.dtype != torch.uint8:
        raise ValueError(f"expected the (T, H, W) uint8 lip ROI of one song, got {tuple(roi.shape)} {roi.dtype}")
    out = []
    for lo, hi in utterance_bounds(roi.shape[0], frame_rate, dur_threshold):
        out.append(encoder({"video": roi[lo:hi].unsqueeze(0), "audio": None, "transform": transform})[0])
    return torch.cat(out, dim=0)
