/*
 * svt_mi355.h — C-ABI of the MI355X-native singing-transcription hot path.
 *
 * This is the drop-in boundary.  The reference is 100 % Python; its "FFI" for this
 * path is the set of torch ops reached from
 *   MIR_ST500/huggingface_interface.py:263-297   (HuggingFaceWav2Vec2.forward / extract_features)
 *   speechbrain/nnet/linear.py:63-76             (Linear.forward, the 20-way frame head)
 *   MIR_ST500/train_audio_ssl.py:41-46,93-100    (logit slicing, sigmoid / argmax per frame)
 *   N20EMv2/audio_visual/fusion.py:192-210       (FusionRCA.forward)
 *   speechbrain/decoders/ctc.py:341-383          (ctc_greedy_decode)
 *   speechbrain/lobes/features.py:126-143        (Fbank.forward)
 * Each entry point below names the interface it replaces.  A maintainer binds them with ctypes
 * (INTEGRATION.md shows the stub); svt_speechbrain_amd/ does exactly that.
 *
 * Conventions
 *   - plain C types only: pointers + sizes, no torch / HIP types in signatures (`stream` is a
 *     hipStream_t passed as void*; NULL = the default stream).
 *   - every *_dev pointer is device memory on the object's device, caller-owned, never mutated when
 *     const.  Host pointers are named *_host.
 *   - all functions return 0 on success or a negative svt_status; svt_last_error() returns a
 *     thread-local text.  Nothing aborts, nothing synchronises the device or allocates inside a
 *     forward call (workspace is caller-supplied), so calls may be captured into a hipGraph.
 *   - one object per device; an object is NOT thread-safe.
 */
#ifndef SVT_MI355_H
#define SVT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVT_ABI_VERSION 1
#define SVT_MAX_CONV_LAYERS 8

typedef enum {
  SVT_OK = 0,
  SVT_ERR_INVALID = -1,     /* bad argument / config / shape              */
  SVT_ERR_KEY = -2,         /* unknown or missing parameter key           */
  SVT_ERR_STATE = -3,       /* call order (e.g. encode before finalize)   */
  SVT_ERR_WORKSPACE = -4,   /* workspace too small                        */
  SVT_ERR_HIP = -5,         /* HIP runtime error (text in last_error)     */
  SVT_ERR_NO_DEVICE = -6    /* no gfx950 device visible                   */
} svt_status;

/* Operand handling of the dense products (conv 1-6 as implicit GEMM, projections, attention, FFN; HF:254-802):
 *   SVT_PREC_FP32    fp32 operands, exact fp32 MFMA (v_mfma_f32_16x16x4_f32): the parity mode, 157 TFLOP/s peak
 *   SVT_PREC_BF16    bf16 operands and activations, fp32 accumulate: the throughput mode
 *   SVT_PREC_BF16X3  fp32 activations / weights in memory; inside the product kernels every operand is cut into bf16
 *                    (hi, lo) and Ah*Wh + Al*Wh + Ah*Wl is accumulated in fp32 on the bf16 matrix pipe (3 MFMAs of 16
 *                    cycles instead of 8 fp32 MFMAs of 32): ~2^-17 relative operand error, fp32 range
 *   SVT_PREC_FP16X3  the same with fp16 pieces (~2^-22 relative; operands must stay below 65504 in magnitude, which
 *                    holds for the fp16-trained wav2vec2 / HuBERT / WavLM checkpoints the wrapper loads) */
typedef enum { SVT_PREC_FP32 = 0, SVT_PREC_BF16 = 1, SVT_PREC_BF16X3 = 2, SVT_PREC_FP16X3 = 3 } svt_precision;
typedef enum { SVT_NORM_GROUP = 0, SVT_NORM_LAYER = 1 } svt_feat_norm;
typedef enum { SVT_F32 = 0 } svt_dtype;

/* Mirrors the HF Wav2Vec2Config / HubertConfig fields the forward reads
 * (huggingface_interface.py:107-124,169-179) plus the wrapper's two flags (:99-100,130). */
typedef struct svt_encoder_config {
  int32_t struct_size;          /* = sizeof(svt_encoder_config) */
  int32_t hidden_size;
  int32_t num_layers;
  int32_t num_heads;
  int32_t intermediate_size;
  int32_t num_conv_layers;
  int32_t conv_dim[SVT_MAX_CONV_LAYERS];
  int32_t conv_kernel[SVT_MAX_CONV_LAYERS];
  int32_t conv_stride[SVT_MAX_CONV_LAYERS];
  int32_t feat_extract_norm;    /* svt_feat_norm */
  int32_t conv_bias;
  int32_t stable_layer_norm;    /* do_stable_layer_norm */
  int32_t feat_proj_layer_norm;
  int32_t pos_conv_kernel;      /* num_conv_pos_embeddings */
  int32_t pos_conv_groups;      /* num_conv_pos_embedding_groups */
  float layer_norm_eps;
  int32_t normalize_wav;        /* wrapper: F.layer_norm(wav, wav.shape) */
  int32_t output_norm;          /* wrapper: F.layer_norm(out, out.shape) */
  int32_t precision;            /* svt_precision: operand type of the MFMA contractions */
  int32_t pos_conv_depth;       /* 1: one weight-normed positional conv + GELU (wav2vec2 / HuBERT); n > 1: data2vec-audio's stack of
                                 * n plain grouped convs, each followed by LayerNorm(no affine) + GELU */
  int32_t rel_pos_buckets;      /* 0: none; > 0: WavLM's gated relative position bias with this many buckets */
  int32_t rel_pos_max_distance; /* WavLM max_bucket_distance */
  int32_t pos_conv_batch_norm;  /* HuBERT conv_pos_batch_norm (HF modeling_hubert.py HubertPositionalConvEmbedding): eval-mode BatchNorm1d in
                                   front of a plain (not weight-normed) positional conv; keys encoder.pos_conv_embed.batch_norm.* */
} svt_encoder_config;

typedef struct svt_encoder svt_encoder;
typedef struct svt_linear svt_linear;
typedef struct svt_rca svt_rca;

/* one decoded frame: what the reference's per-frame loop appends to song_pred
 * (MIR_ST500/train_audio_ssl.py:95-100) */
typedef struct svt_frame {
  float p_on;
  float p_off;
  int32_t octave;
  int32_t pitch_class;
} svt_frame;

const char* svt_last_error(void);
int svt_abi_version(void);
/* The 16-bit operand type of precision code SVT_PREC_BF16 in THIS build of the library: 0 = bf16 (libsvt_mi355.so), 1 = IEEE half
 * (libsvt_mi355_f16.so: the same sources compiled with -DSVT_OPERAND_F16 -- same MFMA rate, three more mantissa bits; serves
 * precision codes 0 and 1 only). */
int svt_operand_type(void);
/* number of visible HIP devices whose arch is gfx950 (0 => the product path must fail loudly) */
int svt_device_count(void);

/* ---- encoder: replaces HuggingFaceWav2Vec2 (huggingface_interface.py:47-297) + the HF model it wraps ---- */
int svt_encoder_create(const svt_encoder_config* cfg, int device, svt_encoder** out);
void svt_encoder_destroy(svt_encoder* e);

/* Optional "global-batch-equivalent" norms for a batch that is ONE SHARD of a larger one (the reference's
 * wrapper normalises over every element of the batch it is given: huggingface_interface.py:289-295).  With a function set, every
 * forward of this encoder calls it twice, in stream order on the forward's stream -- after the waveform moments and after the
 * moments of the encoder output -- with a DEVICE pointer to the (sum, sum of squares) pair in double precision; the function must
 * replace the pair by its sum over all ranks (an all-reduce of 16 bytes: dist.all_reduce on a tensor wrapping the pointer, or
 * ncclAllReduce on `stream`) and return 0.  The statistics are then taken over global_clips clips (the clip count of the whole
 * global batch, equal lengths), and N shards return what one device returns for the whole batch.  Whole-batch norms only
 * (clips_per_norm_group = 0).  fn = NULL restores the per-shard norms, which is what the reference's DataParallel / DDP runs do. */
typedef int (*svt_norm_reduce_fn)(double* sums_dev, int32_t n_doubles, void* stream, void* user);
int svt_encoder_set_norm_reduce(svt_encoder* enc, svt_norm_reduce_fn fn, void* user, int64_t global_clips);
/* copy one parameter by its HF state-dict key (without the wrapper's "model." prefix); both weight-norm
 * spellings of the positional conv are accepted (…conv.weight_g/_v and …parametrizations.weight.original0/1).
 * replaces: Module.load_state_dict (speechbrain/utils/checkpoints.py:69-95, train_audio_ssl.py:232-234) */
int svt_encoder_load_param(svt_encoder* e, const char* key, const void* data_host, int dtype,
                           const int64_t* shape, int ndim);
/* fold weight-norm, pack q/k/v, reorder conv kernels for implicit GEMM, cast to the operand type,
 * upload.  Fails with SVT_ERR_KEY (naming the key) if a required parameter was never loaded. */
int svt_encoder_finalize(svt_encoder* e);
/* read a (possibly folded) parameter back as fp32 by HF key — for state_dict() round trips */
int svt_encoder_get_param(svt_encoder* e, const char* key, void* out_host, int64_t capacity_elems);
int64_t svt_encoder_num_frames(const svt_encoder* e, int64_t n_samples);
int64_t svt_encoder_workspace_bytes(const svt_encoder* e, int32_t batch, int64_t n_samples);
/* replaces: HuggingFaceWav2Vec2.extract_features (:279-297): wav f32 (B,L) -> feats f32 (B,T,D) */
int svt_encoder_forward(svt_encoder* e, const float* wav_dev, int32_t batch, int64_t n_samples,
                        float* feats_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
/* Same, with the wrapper's two whole-tensor layer norms (huggingface_interface.py:289-295) taken over groups of
 * `clips_per_norm_group` consecutive clips instead of the whole batch (0 = whole batch = svt_encoder_forward).  With 1, a batch
 * of B equal-length clips gives what B batch-1 calls give: the reference's evaluation loop (MIR_ST500/train_audio_ssl.py:90
 * asserts batch 1; utterances of a song are 5 s each) run as ONE batch.  batch must be a multiple of the group size. */
int svt_encoder_forward_ex(svt_encoder* e, const float* wav_dev, int32_t batch, int64_t n_samples,
                           float* feats_dev, void* workspace_dev, size_t workspace_bytes, void* stream,
                           int32_t clips_per_norm_group);

/* encoder + whole-batch output norm + frame head (+ per-frame decode) in one call, WITHOUT writing the (B, T, D) features:
 * what AMT.compute_forward computes (MIR_ST500/train_audio_ssl.py:28-48: feats = wav2vec2(wavs); logits = model(feats))
 * followed, when frames_out_dev is given, by the sigmoid / argmax of compute_objectives (:93-100).  The head is linear, so
 * the raw dots x.w are taken in the single pass over the un-normalised encoder output that also sums the two moments of
 * the output norm; a second pass over B*T*n_out values applies (dot - mean * sum(w)) * rstd + bias and decodes.
 * head: an svt_linear with in_features = hidden_size (512 / 768 / 1024) and out_features <= 32 on the same device.
 * logits_out_dev: (B, T, n_out) fp32.  frames_out_dev: B*T svt_frame or NULL (then n_octave / n_class are ignored). */
int svt_encoder_forward_head(svt_encoder* enc, const svt_linear* head, const float* wav_dev, int32_t batch,
                             int64_t n_samples, float* logits_out_dev, svt_frame* frames_out_dev, int32_t n_octave,
                             int32_t n_class, void* workspace_dev, size_t workspace_bytes, void* stream,
                             int32_t clips_per_norm_group);

/* ---- frame head + per-frame decode: replaces speechbrain.nnet.linear.Linear (linear.py:41-76)
 *      and the sigmoid/argmax loop (train_audio_ssl.py:41-46,93-100) ---- */
int svt_linear_create(int32_t in_features, int32_t out_features, int has_bias, int device, svt_linear** out);
void svt_linear_destroy(svt_linear* l);
int svt_linear_load(svt_linear* l, const float* weight_host /*out x in*/, const float* bias_host /*out or NULL*/);
/* y = x W^T + b, fp32 in / fp32 accumulate / fp32 out; rows = product of leading dims */
int svt_linear_forward(svt_linear* l, const float* x_dev, int64_t rows, float* y_dev, void* stream);
/* logits (rows, 2+n_octave+1+n_class+1) -> frames; argmax = first maximum, sigmoid in fp32 */
int svt_decode_frames(const float* logits_dev, int64_t rows, int32_t n_out, int32_t n_octave,
                      int32_t n_class, svt_frame* frames_dev, int device, void* stream);
/* frame2note (reference MIR_ST500/utils.py:82-149; called per song at train_audio_ssl.py:104-108) over a batch of decoded frame
 * sequences ON THE HOST: frames = batch x frames_per_clip records (host memory, as copied back from svt_decode_frames /
 * svt_encoder_forward_head), n_frames = valid frames per clip (NULL: all).  Outputs, per clip b at [b * capacity_per_clip ...]:
 * onset / offset times in seconds (frame_size * frame index, double like the reference's Python floats), pitch (MIDI number =
 * mode of the note's frames + 36), the note's frame range [lo, hi), and n_notes[b].  Where the top pitch count is tied the
 * reference's answer is CPython's max(set(bag), key=bag.count) -- set iteration order -- so pitch is -1 and the caller resolves it
 * from the frames in [lo, hi) (svt_speechbrain_amd/decode.py does).  A one-frame sequence above the onset threshold is the
 * reference's ValueError (max of an empty window): SVT_ERR_INVALID.  No device call is made. */
int svt_frames_to_notes(const svt_frame* frames_host, int32_t batch, int64_t frames_per_clip, const int64_t* n_frames,
                        float onset_thres, float offset_thres, double frame_size, int32_t n_octave, int32_t n_class,
                        double* t_on, double* t_off, int32_t* pitch, int32_t* lo, int32_t* hi, int64_t capacity_per_clip,
                        int64_t* n_notes);

/* ---- RCA fusion: replaces FusionRCA (N20EMv2/audio_visual/fusion.py:186-209) ---- */
int svt_rca_create(int32_t d_model, int32_t nhead, int32_t d_ffn, float alpha, int32_t max_len,
                   int32_t precision, int device, svt_rca** out);
void svt_rca_destroy(svt_rca* r);
/* keys as in the FusionRCA state dict: "fusion.layer1.self_att.att.in_proj_weight", … */
int svt_rca_load_param(svt_rca* r, const char* key, const void* data_host, int dtype,
                       const int64_t* shape, int ndim);
int svt_rca_finalize(svt_rca* r);
int64_t svt_rca_workspace_bytes(const svt_rca* r, int32_t batch, int32_t t_audio);
/* audio (B,T1,D), video (B,T2,D) f32 -> out (B,T1,D) f32; video is truncated / zero-padded to T1 */
int svt_rca_forward(svt_rca* r, const float* audio_dev, int32_t t_audio, const float* video_dev,
                    int32_t t_video, int32_t batch, float* out_dev, void* workspace_dev,
                    size_t workspace_bytes, void* stream);

/* ---- CTC greedy: replaces speechbrain.decoders.ctc.ctc_greedy_decode (ctc.py:341-383) ----
 * probs (B,T,V) f32, rel_lens (B,) f32 (device).  tokens_dev (B,T) i32 receives the collapsed, blank-free
 * ids left-aligned per row; out_lens_dev (B,) i32 their counts.  blank < 0 counts from V. */
int svt_ctc_greedy(const float* probs_dev, int32_t batch, int32_t t, int32_t v, const float* rel_lens_dev,
                   int32_t blank, int32_t* tokens_dev, int32_t* out_lens_dev, int device, void* stream);

/* ---- Fbank: replaces speechbrain.lobes.features.Fbank default chain (features.py:126-143) ----
 * wav (B,L) f32 -> (B, 1+L/hop, n_mels) f32; hamming win, centred, constant pad, power, triangular mel,
 * 10 log10 clamp 1e-10, per-sequence top_db clip. */
int64_t svt_fbank_workspace_bytes(int32_t batch, int64_t n_samples, int32_t n_fft, int32_t hop, int32_t n_mels);
int svt_fbank(const float* wav_dev, int32_t batch, int64_t n_samples, int32_t sample_rate, int32_t n_fft,
              int32_t win_length, int32_t hop_length, int32_t n_mels, float f_min, float f_max, float top_db,
              float* out_dev, void* workspace_dev, size_t workspace_bytes, int device, void* stream);

/* ======================================================================================================================
 * DIAGNOSTIC SURFACE: svt_debug_* and svt_prof_*.  These entry points exist for the unit tests of single kernels, the micro-benchmarks
 * under tools/ and the bench's per-launch timing.  They are the ONE exception to the conventions of the product entry points above:
 * a svt_debug_* call may allocate and free device memory and may synchronise its stream (hipMalloc / hipStreamSynchronize inside), and
 * svt_debug_set changes process-wide kernel selection.  No product entry point (svt_encoder_*, svt_linear_*, svt_rca_*, svt_video_*,
 * svt_decode_frames, svt_ctc_greedy, svt_fbank, svt_deltas, svt_context_window, svt_bce_loss, svt_nll_loss, svt_softmax) calls them, and
 * none of those allocates or synchronises.
 * ====================================================================================================================== */
/* ---- test / micro-benchmark hook: the dense contraction kernel on caller-supplied operands ----
 * C (M,N) = act(A W^T + bias) + resid with A (M,K) and W (N,K) in the operand type of `precision`
 * (fp32, or bf16 bits for SVT_PREC_BF16; the split-operand precisions take fp32 operands), C in the operand type unless out_f32.  a_rpb/a_bstride/a_rstride describe the
 * implicit-conv row addressing (row m starts at (m / a_rpb) * a_bstride + (m % a_rpb) * a_rstride elements);
 * pass a_rpb = M, a_bstride = 0, a_rstride = K for a plain row-major A; ldw = W row pitch (>= K). */
int svt_debug_gemm(int32_t precision, const void* a_dev, const void* w_dev, void* c_dev, const float* bias_dev,
                   const float* resid_dev, int32_t m, int32_t n, int32_t k, int32_t a_rpb, int64_t a_bstride,
                   int64_t a_rstride, int64_t ldw, int32_t act, int32_t out_f32, int device, void* stream);

/* The same hook for the split-operand modes' PAIR-ROW products (csrc/gemm_x3q.hip: both operands pre-cut into 16-bit (hi, lo) pieces,
 * 32 elements per 128-byte line): `a_dev` is the fp32 tensor (a_elems elements, a multiple of 32; rows addressed as above with strides
 * in multiples of 32 elements), converted to pair rows inside the hook; the product runs with the pair-row operand and writes
 * out_kind 0 = fp32 rows, 1 = pair rows, 2 = separate (hi, lo) planes -- 1 and 2 are converted back to fp32 (hi + lo) into c_dev, so
 * the caller always compares an fp32 (M, N) result.  precision = SVT_PREC_BF16X3 / SVT_PREC_FP16X3; act 0 / 1 (exact-erf GELU). */
int svt_debug_gemm_pairs(int32_t precision, const float* a_dev, int64_t a_elems, const float* w_dev, float* c_dev, const float* bias_dev,
                         int32_t m, int32_t n, int32_t k, int32_t a_rpb, int64_t a_bstride, int64_t a_rstride, int32_t act,
                         int32_t out_kind, int device, void* stream, int32_t time_iters, float* ms_out);
/* (time_iters > 0 with ms_out: after the checked launch the product alone is launched time_iters more times between two HIP events;
 *  *ms_out = milliseconds per launch -- tools/x3q_bench.py) */

/* ---- AV-HuBERT lip front-end: replaces SubModel / ResEncoder of N20EMv2/video_only/resnet.py:134-187 (the
 * `feature_extractor_video` of the AV-HuBERT model, hubert.py:344-346): 3-D stem + ResNet-18 trunk (PReLU) + Linear(512,
 * embed_dim), eval mode.  Parameter keys are SubModel.state_dict() keys ("resnet.frontend3D.0.weight", "resnet.trunk.layer1.0.
 * bn1.running_mean", ..., "proj.weight"); num_batches_tracked entries are not needed. ---- */
typedef struct svt_video svt_video;
int svt_video_create(int32_t embed_dim, int32_t precision, int device, svt_video** out);
void svt_video_destroy(svt_video* v);
int svt_video_load_param(svt_video* v, const char* key, const void* data_host, int dtype, const int64_t* shape, int ndim);
/* folds the eval-mode batch norms into the conv weights, re-lays the weights out tap-major / as MFMA fragments, uploads */
int svt_video_finalize(svt_video* v);
int64_t svt_video_workspace_bytes(const svt_video* v, int32_t batch, int32_t t, int32_t h, int32_t w);
/* keep != 0: the caller OWNS the workspace it passes to svt_video_forward -- between two calls nothing else writes it.  The zero halos
 * of the padded stage buffers (which no kernel of the path overwrites) are then written once per (workspace pointer, geometry, stream)
 * instead of on every call: 12 launches, 1.3 GB of stores, 0.25 ms per 16 x 500 frames of 88 x 88.  Default 0: the workspace is scratch
 * and every call rewrites them.  A host-side setting: no device work. */
int svt_video_keep_workspace(svt_video* v, int keep);
/* video (B,1,T,H,W) f32 on the device -> out (B,T,embed_dim) f32 */
int svt_video_forward(svt_video* v, const float* video_dev, int32_t batch, int32_t t, int32_t h, int32_t w, float* out_dev,
                      void* workspace_dev, size_t workspace_bytes, void* stream);
/* the same with a row pitch: out row (b, t) starts at out_dev + (b * t_total + t) * out_ld (out_ld >= embed_dim); zero_left > 0 also zeroes
 * the `zero_left` columns to the LEFT of every row (out_dev - zero_left must be inside the caller's buffer, out_ld >= embed_dim +
 * zero_left; zero_left and out_ld multiples of 4, out_dev - zero_left 16-byte aligned): with out_dev = feats + E, out_ld = 2 E, zero_left = E this writes AV-HuBERT's concat fusion input
 * cat([zeros (absent audio), video], -1) in place (N20EMv2/video_only/hubert.py:700-712) */
int svt_video_forward_ex(svt_video* v, const float* video_dev, int32_t batch, int32_t t, int32_t h, int32_t w, float* out_dev, int64_t out_ld,
                         int32_t zero_left, void* workspace_dev, size_t workspace_bytes, void* stream);
/* ---- the recipe's input side, fused (round 6): replaces transform_eval of N20EMv2/video_only/train_video_ssl.py:445-457 + the
 * .astype(np.float32) of :530-533 in front of the lip front-end.  roi_dev: (B, T, h_in, w_in) uint8 gray frames as np.load returns them,
 * on the device.  Every pixel u becomes float32(((double(u) - sub0) / div0 - mean) / std) -- numpy's float64 arithmetic rounded once,
 * bit-identical -- of the centre crop_h x crop_w window (offsets int(round(h_in - crop_h) / 2.), utils.py:79-83).  The recipes use
 * {0.0, 255.0, 0.421, 0.165, 88, 88}.  Workspace: svt_video_workspace_bytes(v, batch, t, crop_h, crop_w).  out_ld / zero_left as above. */
typedef struct svt_video_transform { double sub0, div0, mean, std; int32_t crop_h, crop_w; } svt_video_transform;
int svt_video_forward_u8(svt_video* v, const uint8_t* roi_dev, int32_t batch, int32_t t, int32_t h_in, int32_t w_in,
                         const svt_video_transform* tf, float* out_dev, int64_t out_ld, int32_t zero_left, void* workspace_dev,
                         size_t workspace_bytes, void* stream);

/* ---- Fbank add-ons: replace speechbrain.processing.features.Deltas (features.py:788-850; replicate padding, kernel
 * -n..n, denominator n(n+1)(2n+1)/3) and ContextWindow (:853-940; out[..., c*ctx + j], zero padding).  x rows have pitch ldx
 * (so a delta can be written next to its source inside a wider feature tensor). ---- */
int svt_deltas(const float* x_dev, int64_t ldx, int32_t batch, int32_t t, int32_t c, int32_t window_length, float* out_dev,
               int64_t ldo, int device, void* stream);
int svt_context_window(const float* x_dev, int32_t batch, int32_t t, int32_t c, int32_t left_frames, int32_t right_frames,
                       float* out_dev, int device, void* stream);

/* ---- validation losses: replace speechbrain.nnet.losses.bce_loss / nll_loss (losses.py:402-519) over
 * compute_masked_loss (:624-684), truncate (:594-621) and length_to_mask (dataio/dataio.py:661-706); forward only ----
 * logits (B,t_pred) f32, targets (B,t_tgt) f32; the longer of the two is truncated when |t_pred - t_tgt| <=
 * allowed_len_diff, otherwise SVT_ERR_INVALID with the reference's message.  rel_len (B,) f32 relative lengths or
 * NULL; pos_weight: device pointer to ONE float or NULL.  reduction: 0 mean, 1 batchmean, 2 batch ((B,) outputs),
 * 3 none ((B,T) masked per-frame losses).  workspace: batch*24+8 bytes of device memory. */
int svt_bce_loss(const float* logits_dev, int64_t batch, int64_t t_pred, const float* targets_dev, int64_t t_tgt,
                 const float* rel_len_dev, const float* pos_weight_dev, int32_t allowed_len_diff, int32_t reduction,
                 float* out_dev, void* workspace_dev, size_t workspace_bytes, int device, void* stream);
/* log_probs (B,t_pred,n_class) f32, targets (B,t_tgt) i64 (-100 = ignored, as torch.nn.functional.nll_loss).  After the
 * call the int32 at workspace + batch*24 is non-zero if a target was outside [0, n_class). */
int svt_nll_loss(const float* log_probs_dev, int64_t batch, int64_t t_pred, int32_t n_class, const int64_t* targets_dev,
                 int64_t t_tgt, const float* rel_len_dev, float label_smoothing, int32_t allowed_len_diff,
                 int32_t reduction, float* out_dev, void* workspace_dev, size_t workspace_bytes, int device, void* stream);
/* y = softmax(x) or log_softmax(x) over the last axis of (rows, n): replaces speechbrain.nnet.activations.Softmax
 * (activations.py:22-75) as the recipes use it (hparams log_softmax, apply_log=True) */
int svt_softmax(const float* x_dev, int64_t rows, int32_t n, int32_t apply_log, float* y_dev, int device, void* stream);

/* Fused attention kernel alone (bf16, head_dim 64 or 128): o[b,t,h*dh+d] = softmax(scale q k^T) v with q rows at
 * q + (b*t_len + t)*ldq + h*dh, k / v rows likewise with ldkv, o with ldo (element strides).  Replaces the eager
 * attention of transformers' Wav2Vec2Attention / torch.nn.MultiheadAttention inside the encoder and the RCA layers
 * (N20EMv2/audio_visual/fusion.py:102-117); exposed for the parity tests and tools/attn_bench.py. */
int svt_debug_attention(int32_t precision, const void* q, const void* k, const void* v, void* o, int32_t batch, int32_t t,
                        int32_t heads, int32_t head_dim, int64_t ldq, int64_t ldkv, int64_t ldo, float scale, int device,
                        void* stream);
/* Diagnostics switches of the contraction kernels (tools/gemm_bench.py, tools/gemm_trace.py; never needed in production):
 * key 0 = kernel ablation variant, 1 = force the tile height (64/128/192/256), 2 = force the one-tile (2) / persistent (4)
 * scheduler, 3 = ablation variant while tracing, 5 = retired (the fused out-projection + LayerNorm kernel of rounds 1-2), 6 = small-problem kernel
 * (gemm_skinny.hip) on/off, 7 = its eligibility threshold in 128x256 tiles, 8 = 8-wave fused-attention workgroups on/off,
 * 9 = bf16 (1) or fp32 (0) convolution output in front of the conv-stack LayerNorm in bf16 mode, 10 = whole-head fused
 * attention kernel (K / V of a head resident in LDS; measured slower, off by default), 11 = LDS-DMA split-operand GEMM
 * kernel on/off (off: the register-staged one), 12 = svt_debug_gemm keeps the split copy of its weight between calls,
 * 13 = page-guarded device allocations (see svt_debug_alloc), 19 = split-operand modes keep product operands as pair rows written by
 * their producers (1, default) or as fp32 cut inside the product kernels (0: the round-2/3 path, A/B), 20 = (hi, lo) LayerNorm with two
 * rows per wave (1, default) or one (0), 21 = fused attention variant (0 = staggered wave groups, 1 = the round-3 lockstep kernel),
 * 22 = conv layer 0 on the matrix pipe in the 16-bit modes (1, default) or on the vector ALU (0), 23 = stage 1 of the lip front-end on
 * the frame-resident direct convolution (1, default; conv3x3_c64.hip) or on the GEMM kernels (0), 24 = query: returns the number of
 * direct-convolution launches of this process so far (value ignored), 25 = timing-ablation bits of the direct convolution (DIAG builds),
 * 26 = stem + max-pool of the lip front-end as one persistent kernel (1, default) or as two kernels (0), 27 = stage 2's 1x1 stride-2
 * downsample inside conv1's product (1, default) or as its own product (0), 28 = two-slot schedule of gemm_pps_kernel (0, default: four
 * slots), 29 = gemm_p1w_kernel (one wave per SIMD) for the 16-bit products it measured faster on (1, default), everywhere (2) or never (0),
 * 30 = gemm_p1x_kernel (the same loop for the split-operand products on pair rows: 1) or gemm_x3q_kernel (0, default: same bits, same speed),
 * 31 = query: returns the number of device buffers this process has FREED so far (value ignored; uploads and re-uploads must not free:
 * tests/test_gpu_uploads.py), 32 = the tickets of the three ordered cross-workgroup sums as acquire-release atomics (1) or relaxed ones behind
 * write-through stores (0, default: kernels.hip, last_workgroup; same bits, tests/test_gpu_statistics.py), 34 = the persistent 16-bit GEMM
 * kernels' tile walk: -1 (default) chosen per launch, 0 = n fastest, n > 0 = panels of n tile rows (common.h, tile_walk; same bits), 35 = the kernel-3 convolutions' K slabs tap-minor (1, default: the frame two
 * neighbouring output rows share is re-read out of L2) or tap-major (0) on gemm_p1w_kernel, 36 = FFN-2 of a small batch (<= 2048 rows) as a K-split small GEMM whose partial
 * products the following LayerNorm adds (1, default) or as one product (0), 33 = the small-problem GEMM uses 32 x 32 tiles while its 64 x 64 tiling has at most this many workgroups
 * (96, default; swept on the one-utterance forward: 150 / 200 / 1000 are 1.5-4.5 % slower).
 * Returns 0 (keys 24, 31: the count), SVT_ERR_INVALID for an unknown key. */
int svt_debug_set(int key, int value);

/* ---- measurement hook: HIP-event timing of the dominant kernel on the stream it runs on ----
 * When enabled, forward calls bracket every launch of the dense contraction kernel with hipEvents on
 * the launch stream; svt_prof_read returns accumulated launches / milliseconds / algorithmic flops. */
/* on = 0: off; 1: a HIP-event pair around every dense-contraction / attention launch; n > 1: around every n-th one */
int svt_prof_enable(int on);
int svt_prof_reset(void);
/* kind: 0 = the dominant kernel family (svt::gemm_pps_kernel / gemm_pers_kernel / gemm_pp8_kernel: every large bf16
 * dense contraction; in the split-operand modes the split form of svt::gemm_kernel), 1 = the other dense contraction
 * kernels (exact-fp32 / small-shape GEMM), 2 = fused attention */
int svt_prof_read(int kind, int64_t* launches, double* total_ms, double* total_flops, double* total_bytes);
/* Clock stamps for a sustained-rate measurement (bench.py): enqueues a tiny kernel on `stream` that stores, per XCD x
 * (0..7), out_dev[2x] = s_memtime (shader-clock ticks) and out_dev[2x + 1] = s_memrealtime (100 MHz ticks) as int64.
 * Two calls around a region give the clock the chip HELD over it: d(memtime) / d(memrealtime) * 100 MHz, per XCD.
 * out_dev: 16 x int64, zeroed by the caller (an XCD that ran no block of the kernel leaves its pair untouched). */
int svt_debug_clock(int64_t* out_dev, int device, void* stream);
/* Diagnostics: a device buffer from the library's own allocator.  With svt_debug_set(13, 1 | 2) every allocation of the library
 * (these included) is its own mapping between two unmapped granules of address space, flush against the end (1) or the start (2)
 * of the mapping: an out-of-bounds access by a kernel on that side faults instead of landing in a neighbour
 * (tests/test_gpu_guard.py runs the forward passes with weights, workspace and inputs placed this way). */
int svt_debug_alloc(void** out, size_t bytes, int device);
/* Diagnostics (tools/determinism_stress.py): byte offsets of the regions svt_encoder_forward* carves out of the caller's workspace for
 * (batch, n_samples), in carve order -- 0 moments + ordered-sum scratch, 1 conv0 coefficients, 2 conv0 tables, 3 / 4 conv activations
 * (ping-pong), 5 fp32 conv output (layer-norm extractors), 6 projection input, 7 hF, 8 preF, 9 layer input, 10 fp32 residual, 11 low half
 * of the residual, 12 / 13 positional-conv operand / result, 14 QKV, 15 / 16 scores / probabilities (non-fused attention), 17 V transposed,
 * 18 QKV planes (split modes), 19 attention output, 20 FFN hidden, 21 gate, 22 relative-position bias, 23 head dots, 24 total.  A region the
 * configuration does not use has offset -1.  Writes min(n, 25) entries; returns the number written or a negative svt_status. */
int svt_debug_encoder_layout(const svt_encoder* e, int32_t batch, int64_t n_samples, int64_t* offsets, int n);
int svt_debug_free(void* p, int device);

#ifdef __cplusplus
}
#endif
#endif /* SVT_MI355_H */
