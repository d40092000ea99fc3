"""svt_speechbrain_amd — MI355X-native singing-transcription forward path (drop-in for the hot path of
guxm2021/SVT_SpeechBrain; see DESIGN.md / INTEGRATION.md).  Importing the package never touches the GPU."""
from .config import EncoderConfig, PRESETS, config_from_source  # noqa: F401
from .huggingface_interface import HuggingFaceWav2Vec2  # noqa: F401
from .linear import Linear  # noqa: F401
from .fusion import FusionRCA  # noqa: F401
from .features import Fbank  # noqa: F401
from .decode import decode_frames, frame2note, frames2note, frames2note_batch, frames_to_info, ctc_greedy_decode, filter_ctc_output  # noqa: F401
from .amt import AMTForward  # noqa: F401
from . import losses  # noqa: F401
from . import checkpoints  # noqa: F401
from .checkpoints import Checkpointer  # noqa: F401
from .losses import bce_loss, nll_loss, Softmax  # noqa: F401
from . import video  # noqa: F401
from . import dataio, scoring  # noqa: F401
from .video import FairseqAVHubertPretrain, EvalTransform  # noqa: F401
from .song import (SongTranscriber, utterance_bounds, save_song_features, feature_path, song_video_features,  # noqa: F401
                   save_song_video_features, video_feature_path)

__all__ = ["EncoderConfig", "PRESETS", "config_from_source", "HuggingFaceWav2Vec2", "Linear", "FusionRCA", "Fbank",
           "decode_frames", "frame2note", "frames2note", "frames2note_batch", "frames_to_info", "ctc_greedy_decode", "filter_ctc_output", "AMTForward",
           "SongTranscriber", "utterance_bounds", "save_song_features", "feature_path", "song_video_features",
           "save_song_video_features", "video_feature_path", "EvalTransform", "FairseqAVHubertPretrain"]
