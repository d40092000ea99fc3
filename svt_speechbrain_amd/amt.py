"""The recipe-level hot path: ``AMT.compute_forward`` (reference ``MIR_ST500/train_audio_ssl.py:28-48``,
twin ``N20EMv2/audio_visual/train_rca_av.py:28-51``) and the decode half of ``compute_objectives``
(``train_audio_ssl.py:85-107``), without the SpeechBrain trainer around them.

``AMTForward`` takes the same ``modules`` mapping a recipe yaml builds (``wav2vec2`` + ``model`` for the audio
recipes, ``fusion`` + ``model`` for the audio-visual one) and returns the reference's 5-tuple."""
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
        self.fuse_tail = True   # False: encoder and head as two module calls (the features are materialised)
        self.song_pred: list = []

    def _get(self, name):
        m = self.modules
        return m[name] if isinstance(m, dict) or hasattr(m, "__getitem__") else getattr(m, name)

    def compute_forward(self, wavs: torch.Tensor, wav_lens: Optional[torch.Tensor] = None, videos: Optional[torch.Tensor] = None):
        """-> (onset_logits, offset_logits, pitch_octave_logits, pitch_class_logits, wav_lens)."""
        if videos is not None:
            feats = self._get("fusion")(wavs, videos)
            logits = self._get("model")(feats)
        else:
            enc, head = self._get("wav2vec2"), self._get("model")
            if self.fuse_tail and hasattr(enc, "forward_head") and enc.can_fuse_head(head) and wavs.is_cuda \
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
