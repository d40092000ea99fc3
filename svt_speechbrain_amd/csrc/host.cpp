// Host-only routines of the C-ABI (see host.h): no HIP call, no device pointer is dereferenced here.
#include "host.h"

#include <algorithm>

namespace svt {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

int load_param_into(ParamMap& m, const char* key, const void* data, int dtype, const int64_t* shape, int ndim) {
  if (!key || !data || (ndim > 0 && !shape) || ndim < 0 || ndim > 6) { set_error("load_param: bad argument"); return SVT_ERR_INVALID; }
  if (dtype != SVT_F32) { set_error("load_param: only SVT_F32 host tensors are accepted"); return SVT_ERR_INVALID; }
  Param p;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) {   // no negative extent, no product beyond 2^40 elements (also keeps the multiplication from overflowing)
    if (shape[i] < 0 || (shape[i] > 0 && n > ((int64_t)1 << 40) / shape[i])) { set_error("load_param: bad shape"); return SVT_ERR_INVALID; }
    n *= shape[i];
  }
  if (ndim > 0) p.shape.assign(shape, shape + ndim);
  p.v.assign((const float*)data, (const float*)data + n);
  m[key] = std::move(p);
  return SVT_OK;
}

const Param* find(const ParamMap& m, const std::string& k) {
  auto it = m.find(k);
  return it == m.end() ? nullptr : &it->second;
}

int need(const ParamMap& m, const std::string& k, std::vector<int64_t> shape, const Param** out) {
  const Param* p = find(m, k);
  if (!p) { set_error("missing parameter: " + k); return SVT_ERR_KEY; }
  if (p->shape != shape) {
    std::string s = "parameter " + k + " has shape (";
    for (auto d : p->shape) s += std::to_string(d) + ",";
    s += ") expected (";
    for (auto d : shape) s += std::to_string(d) + ",";
    set_error(s + ")");
    return SVT_ERR_INVALID;
  }
  *out = p;
  return SVT_OK;
}

int validate_cfg(const svt_encoder_config& c) {
  if (c.struct_size != (int32_t)sizeof(svt_encoder_config)) { set_error("svt_encoder_config: struct_size mismatch (ABI)"); return SVT_ERR_INVALID; }
  if (c.num_conv_layers < 0 || c.num_conv_layers > SVT_MAX_CONV_LAYERS) { set_error("num_conv_layers out of range"); return SVT_ERR_INVALID; }
  if (c.num_conv_layers == 0) {
    // features-in mode (AV-HuBERT video branch): the input is a (B, T, conv_dim[0]) feature tensor, the path starts at
    // the feature projection
    if (c.conv_dim[0] < 8 || c.conv_dim[0] % 8) { set_error("features-in mode: conv_dim[0] (the feature width) must be a multiple of 8"); return SVT_ERR_INVALID; }
    if (c.normalize_wav) { set_error("features-in mode: normalize_wav does not apply"); return SVT_ERR_INVALID; }
  } else if (c.conv_kernel[0] != 10) { set_error("conv layer 0 must have kernel 10 (wav2vec2/HuBERT geometry)"); return SVT_ERR_INVALID; }
  if (c.num_conv_layers > 0 && (c.conv_stride[0] > 5 || c.conv_stride[0] < 1)) { set_error("conv layer 0 stride must be 1..5"); return SVT_ERR_INVALID; }
  for (int i = 0; i < c.num_conv_layers; ++i) {
    if (c.conv_dim[i] % 8 || c.conv_dim[i] < 8) { set_error("conv_dim must be a multiple of 8"); return SVT_ERR_INVALID; }
    if (c.conv_kernel[i] < 1 || c.conv_stride[i] < 1) { set_error("bad conv geometry"); return SVT_ERR_INVALID; }
  }
  if (c.num_conv_layers > 0 && c.conv_dim[0] > 512) { set_error("conv_dim[0] > 512 unsupported"); return SVT_ERR_INVALID; }
  if (c.hidden_size < 8 || c.num_heads < 1 || c.pos_conv_groups < 1 || c.intermediate_size < 8 || c.num_layers < 0 || c.pos_conv_kernel < 1) {
    set_error("hidden_size / num_heads / pos_conv_groups / intermediate_size / num_layers / pos_conv_kernel out of range"); return SVT_ERR_INVALID; }
  if (c.hidden_size % c.num_heads) { set_error("hidden_size % num_heads != 0"); return SVT_ERR_INVALID; }
  const int dh = c.hidden_size / c.num_heads;
  if (dh % 8) { set_error("head_dim must be a multiple of 8"); return SVT_ERR_INVALID; }
  if (c.hidden_size % c.pos_conv_groups || (c.hidden_size / c.pos_conv_groups) % 8) { set_error("hidden_size/pos_conv_groups must be a multiple of 8"); return SVT_ERR_INVALID; }
  if (c.intermediate_size % 8 || c.hidden_size % 8) { set_error("sizes must be multiples of 8"); return SVT_ERR_INVALID; }
  if (c.feat_extract_norm != SVT_NORM_GROUP && c.feat_extract_norm != SVT_NORM_LAYER) { set_error("feat_extract_norm"); return SVT_ERR_INVALID; }
  if (!valid_precision(c.precision)) { set_error("precision"); return SVT_ERR_INVALID; }
  if (c.pos_conv_depth < 1 || c.pos_conv_depth > 16) { set_error("pos_conv_depth must be 1..16"); return SVT_ERR_INVALID; }
  if (c.pos_conv_batch_norm && c.pos_conv_depth != 1) { set_error("pos_conv_batch_norm applies to the single positional conv only"); return SVT_ERR_INVALID; }
  if (c.rel_pos_buckets < 0 || c.rel_pos_buckets % 4 || (c.rel_pos_buckets > 0 && c.rel_pos_max_distance <= c.rel_pos_buckets / 4)) {
    set_error("rel_pos_buckets must be a multiple of 4 and rel_pos_max_distance > rel_pos_buckets / 4"); return SVT_ERR_INVALID; }
  return SVT_OK;
}

}  // namespace svt

using namespace svt;

extern "C" {

const char* svt_last_error(void) { return svt::g_err.c_str(); }

// frame2note (reference MIR_ST500/utils.py:82-149) over a batch of decoded frame sequences, on the HOST (no device call).
// Same scan as the reference's loop: float32 comparisons against the thresholds, onset = above threshold AND equal to the
// maximum of onset[i-3 : min(i+4, N-1)] (the last frame is never inside a window), times = frame_size * i in double.
// The pitch of a note is the mode of its bag; where the top count is tied the reference's answer depends on CPython's set
// iteration order, so the note is returned with pitch = -1 and its frame range [lo, hi) and the caller resolves it.
int svt_frames_to_notes(const svt_frame* frames, int32_t batch, int64_t frames_per_clip, const int64_t* n_frames,
                        float onset_thres, float offset_thres, double frame_size, int32_t n_octave, int32_t n_class,
                        double* t_on, double* t_off, int32_t* pitch, int32_t* lo, int32_t* hi, int64_t capacity_per_clip,
                        int64_t* n_notes) {
  if (!frames || !t_on || !t_off || !pitch || !lo || !hi || !n_notes || batch < 0 || frames_per_clip < 0) {
    set_error("svt_frames_to_notes: bad argument"); return SVT_ERR_INVALID; }
  if (n_octave < 1 || n_class < 1 || (long)n_octave * n_class + n_class > 4096) { set_error("svt_frames_to_notes: bad class counts"); return SVT_ERR_INVALID; }
  std::vector<int> counts((size_t)n_octave * n_class + n_class + 1);
  for (int32_t b = 0; b < batch; ++b) {
    const svt_frame* f = frames + (int64_t)b * frames_per_clip;
    const int64_t n = n_frames ? n_frames[b] : frames_per_clip;
    if (n < 0 || n > frames_per_clip) { set_error("svt_frames_to_notes: n_frames out of range"); return SVT_ERR_INVALID; }
    int64_t k = 0;
    const int64_t base = (int64_t)b * capacity_per_clip;
    bool open = false;
    int64_t on_i = 0, bag_n = 0;
    auto emit = [&](int64_t close_frame, int64_t hi_frame) -> int {   // the open note [on_i, hi_frame) closes at time frame_size * close_frame
      if (k >= capacity_per_clip) { set_error("svt_frames_to_notes: more notes than capacity_per_clip"); return SVT_ERR_INVALID; }
      int top = 0, arg = 0, ties = 0;
      for (size_t v = 0; v < counts.size(); ++v) {
        if (counts[v] > top) { top = counts[v]; arg = (int)v; ties = 1; }
        else if (counts[v] == top && top > 0) ++ties;
      }
      t_on[base + k] = frame_size * (double)on_i;
      t_off[base + k] = frame_size * (double)close_frame;
      pitch[base + k] = ties > 1 ? -1 : arg + 36;
      lo[base + k] = (int32_t)on_i;
      hi[base + k] = (int32_t)hi_frame;
      ++k;
      return 0;
    };
    for (int64_t i = 0; i < n; ++i) {
      const float p_on = f[i].p_on;
      bool is_on = false;
      if (p_on >= onset_thres) {
        const int64_t w0 = i - 3 > 0 ? i - 3 : 0, w1 = i + 4 < n - 1 ? i + 4 : n - 1;
        if (w1 <= w0) { set_error("frame2note: max() of an empty onset window (a one-frame sequence above the onset threshold), as in the reference"); return SVT_ERR_INVALID; }
        float m = f[w0].p_on;
        for (int64_t j = w0 + 1; j < w1; ++j) m = f[j].p_on > m ? f[j].p_on : m;
        is_on = p_on == m;
      }
      if (is_on) {
        if (open && bag_n) { if (int r = emit(i, i)) return r; }
        open = true; on_i = i; bag_n = 0;
        std::fill(counts.begin(), counts.end(), 0);
      } else if (f[i].p_off >= offset_thres) {
        if (open) {
          if (bag_n) { if (int r = emit(i, i)) return r; }
          open = false; bag_n = 0;
        }
      }
      if (open && f[i].octave != n_octave && f[i].pitch_class != n_class) {
        const long v = (long)f[i].octave * n_class + f[i].pitch_class;
        if (v < 0 || v >= (long)counts.size()) { set_error("svt_frames_to_notes: frame class out of range"); return SVT_ERR_INVALID; }
        ++counts[v];
        ++bag_n;
      }
    }
    if (open && bag_n) { if (int r = emit(n - 1, n)) return r; }
    n_notes[b] = k;
  }
  return SVT_OK;
}

}  // extern "C"
