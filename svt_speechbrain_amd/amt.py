"""The recipe-level hot path: ``AMT.compute_forward`` (reference ``MIR_ST500/train_audio_ssl.py:28-48``,
twin ``N20EMv2/audio_visual/train_rca_av.py:28-51``) and the decode half of ``compute_objectives``
(``train_audio_ssl.py:85-107``), without the SpeechBrain trainer around them.

``AMTForward`` takes the same ``modules`` mapping a recipe yaml builds -- ``wav2vec2`` + ``model`` for the audio recipes
(``MIR_ST500/hparams/train_audio_ssl.yaml`` ``modules:``), ``fusion`` + ``head`` for the audio-visual one
(``N20EMv2/audio_visual/hparams/train_rca_av.yaml`` ``modules:``, used at ``train_rca_av.py:39,44``) -- and returns the
reference's 5-tuple."""
from __future__ import annotations

from typing import List, Optional

import torch

from .decode import decode_frames, frame2note, frames_to_info, frames2note, frames2note_batch


class AMTForward:
    def __init__(self, modules, pitch_octave_num: int = 4, pitch_class_num: int = 12, onset_threshold: float = 0.4,
                 offset_threshold: float = 0.5, frame_rate: float = 49.8):
        self.modules = modules
        self.pitch_octave_num = pitch_octave_num
        self.pitch_class_num = pitch_class_num
        self.onset_threshold = onset_threshold
        self.offset_threshold = offset_threshold
        self.frame_rate = frame_rate
        # None (default): the fused tail (out-norm + head + decode behind the encoder, the features never written) in the
        # throughput precisions only; the parity-grade precisions (fp32 / fp16x3 / bf16x3) keep the reference's order of
        # operations -- normalise, THEN the linear head -- because the fused form (x.w - mean * sum(w)) * rstd + b re-associates
        # it (2e-4 on the goldens, more by cancellation when |mean| >> std).  True / False force it either way.
        self.fuse_tail = None
        self.song_pred: list = []

    def _has(self, name) -> bool:
        m = self.modules
        if isinstance(m, dict) or hasattr(m, "__contains__"):
            try:
                return name in m
            except TypeError:
                pass
        return hasattr(m, name)

    def _get(self, name):
        m = self.modules
        return m[name] if isinstance(m, dict) or hasattr(m, "__getitem__") else getattr(m, name)

    def _head(self, audio_visual: bool):
        """The frame head under the name the recipe's yaml gives it: ``head`` in the audio-visual recipe
        (train_rca_av.py:44), ``model`` in the audio-only ones (train_audio_ssl.py:39); either is accepted in both."""
        for name in (("head", "model") if audio_visual else ("model", "head")):
            if self._has(name):
                return self._get(name)
        raise KeyError("modules has neither 'head' nor 'model' (the 20-way frame head)")

    def compute_forward(self, wavs: torch.Tensor, wav_lens: Optional[torch.Tensor] = None, videos: Optional[torch.Tensor] = None):
        """-> (onset_logits, offset_logits, pitch_octave_logits, pitch_class_logits, wav_lens)."""
        if videos is not None:
            # train_rca_av.py:28-51: `wavs` / `videos` are the pre-extracted audio / video FEATURES of the batch
            feats = self._get("fusion")(wavs, videos)
            logits = self._head(True)(feats)
        else:
            enc, head = self._get("wav2vec2"), self._head(False)
            fuse = self.fuse_tail if self.fuse_tail is not None else getattr(enc, "precision", None) in ("bf16", "fp16")
            if fuse and hasattr(enc, "forward_head") and enc.can_fuse_head(head) and wavs.is_cuda \
                    and enc.config.hidden_size == head.w.in_features:
                # the features are not an output of compute_forward: out-norm + head in one pass behind the encoder
                logits = enc.forward_head(wavs, head)
            else:
                logits = head(enc(wavs))
        self.last_logits = logits
        o = self.pitch_octave_num
        pitch_out = logits[:, :, 2:]
        return logits[:, :, 0], logits[:, :, 1], pitch_out[:, :, 0:o + 1], pitch_out[:, :, o + 1:], wav_lens

    def decode_utterance(self, logits: torch.Tensor, last_of_song: bool) -> Optional[List[list]]:
        """Append one utterance's frames to the running song (reference asserts batch == 1, :90) and, at the
        last utterance, emit the notes of the whole song."""
        if logits.shape[0] != 1:
            raise AssertionError("batch_size must be 1 during evaluation")
        frames = decode_frames(logits[0], self.pitch_octave_num, self.pitch_class_num)
        self.song_pred.extend(frames_to_info(frames))
        if not last_of_song:
            return None
        notes = frame2note(self.song_pred, self.onset_threshold, self.offset_threshold, 1 / self.frame_rate)
        self.song_pred = []
        return notes

    def transcribe_batch(self, logits: torch.Tensor) -> List[List[list]]:
        """Throughput path: every clip of a batch is its own song (one kernel + one D2H for the batch)."""
        frames = decode_frames(logits, self.pitch_octave_num, self.pitch_class_num)
        return frames2note_batch(frames, self.onset_threshold, self.offset_threshold, 1 / self.frame_rate,
                                 pitch_octave_num=self.pitch_octave_num, pitch_class_num=self.pitch_class_num)
