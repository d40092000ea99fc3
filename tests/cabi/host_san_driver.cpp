// CPU driver of the host-only routines of the C-ABI (svt_speechbrain_amd/csrc/host.cpp) for the sanitizer build (`make san`):
// AddressSanitizer + UndefinedBehaviorSanitizer, -fno-sanitize-recover: any finding aborts with a non-zero exit code.
//   host_san_test notes <frames.bin> <B> <T> <onset> <offset> <frame_size> [capacity]   -> one line per note: "b t_on t_off pitch lo hi"
//                                                          (frames.bin: B * T records of {f32 p_on, f32 p_off, i32 octave, i32 class})
//   host_san_test hostile                                   -> every refusal path of the four routines; prints "hostile ok"
#include "../../svt_speechbrain_amd/csrc/host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace svt;

#define EXPECT(cond)                                                                     \
  do {                                                                                   \
    if (!(cond)) { std::fprintf(stderr, "%s:%d: EXPECT(%s) failed: %s\n", __FILE__, __LINE__, #cond, svt_last_error()); return 1; } \
  } while (0)

static int run_notes(int argc, char** argv) {
  if (argc < 8) return 2;
  const int B = std::atoi(argv[3]);
  const long T = std::atol(argv[4]);
  const float on = (float)std::atof(argv[5]), off = (float)std::atof(argv[6]);
  const double fs = std::atof(argv[7]);
  const long cap = argc > 8 ? std::atol(argv[8]) : (T > 0 ? T : 1);
  std::vector<svt_frame> fr((size_t)B * T);
  FILE* f = std::fopen(argv[2], "rb");
  if (!f) return 2;
  const size_t got = fr.empty() ? 0 : std::fread(fr.data(), sizeof(svt_frame), fr.size(), f);
  std::fclose(f);
  if (got != fr.size()) return 2;
  // exact-size output arrays: an off-by-one write in the scan lands in a red zone
  std::vector<double> t_on((size_t)B * cap), t_off((size_t)B * cap);
  std::vector<int32_t> pitch((size_t)B * cap), lo((size_t)B * cap), hi((size_t)B * cap);
  std::vector<int64_t> n((size_t)B);
  svt_frame dummy{};
  const int rc = svt_frames_to_notes(fr.empty() ? &dummy : fr.data(), B, T, nullptr, on, off, fs, 4, 12, t_on.data(), t_off.data(), pitch.data(),
                                     lo.data(), hi.data(), cap, n.data());
  if (rc != SVT_OK) { std::printf("error %d %s\n", rc, svt_last_error()); return 0; }
  for (int b = 0; b < B; ++b)
    for (long k = 0; k < n[b]; ++k)
      std::printf("%d %.17g %.17g %d %d %d\n", b, t_on[b * cap + k], t_off[b * cap + k], pitch[b * cap + k], lo[b * cap + k], hi[b * cap + k]);
  return 0;
}

static svt_encoder_config good_cfg() {
  svt_encoder_config c;
  std::memset(&c, 0, sizeof c);
  c.struct_size = (int32_t)sizeof c;
  c.num_conv_layers = 7;
  const int k[7] = {10, 3, 3, 3, 3, 2, 2}, st[7] = {5, 2, 2, 2, 2, 2, 2};
  for (int i = 0; i < 7; ++i) { c.conv_dim[i] = 512; c.conv_kernel[i] = k[i]; c.conv_stride[i] = st[i]; }
  c.hidden_size = 768; c.num_layers = 12; c.num_heads = 12; c.intermediate_size = 3072;
  c.pos_conv_kernel = 128; c.pos_conv_groups = 16; c.pos_conv_depth = 1;
  c.feat_extract_norm = SVT_NORM_GROUP; c.precision = SVT_PREC_FP32;
  return c;
}

static int run_hostile() {
  // ---- parameter intake
  ParamMap m;
  const float data[24] = {1, 2, 3, 4, 5, 6};
  const int64_t s23[2] = {2, 3}, neg[2] = {-2, 3}, huge[3] = {(int64_t)1 << 30, (int64_t)1 << 30, 4}, s0[1] = {0};
  EXPECT(load_param_into(m, "w", data, SVT_F32, s23, 2) == SVT_OK && m["w"].v.size() == 6 && m["w"].numel() == 6);
  EXPECT(load_param_into(m, "w", data, SVT_F32, s23, 2) == SVT_OK && m.size() == 1);         // same key again: replaced
  EXPECT(load_param_into(m, "scalar", data, SVT_F32, nullptr, 0) == SVT_OK && m["scalar"].v.size() == 1 && m["scalar"].shape.empty());
  EXPECT(load_param_into(m, "empty", data, SVT_F32, s0, 1) == SVT_OK && m["empty"].v.empty());
  EXPECT(load_param_into(m, nullptr, data, SVT_F32, s23, 2) == SVT_ERR_INVALID);
  EXPECT(load_param_into(m, "x", nullptr, SVT_F32, s23, 2) == SVT_ERR_INVALID);
  EXPECT(load_param_into(m, "x", data, SVT_F32, nullptr, 2) == SVT_ERR_INVALID);
  EXPECT(load_param_into(m, "x", data, SVT_F32, s23, -1) == SVT_ERR_INVALID);
  EXPECT(load_param_into(m, "x", data, SVT_F32, s23, 7) == SVT_ERR_INVALID);
  EXPECT(load_param_into(m, "x", data, SVT_F32 + 1, s23, 2) == SVT_ERR_INVALID);
  EXPECT(load_param_into(m, "x", data, SVT_F32, neg, 2) == SVT_ERR_INVALID);
  EXPECT(load_param_into(m, "x", data, SVT_F32, huge, 3) == SVT_ERR_INVALID);                 // would overflow / exhaust memory
  const Param* p = nullptr;
  EXPECT(need(m, "w", {2, 3}, &p) == SVT_OK && p && p->v[5] == 6.f);
  EXPECT(need(m, "w", {3, 2}, &p) == SVT_ERR_INVALID && std::string(svt_last_error()).find("expected (3,2,)") != std::string::npos);
  EXPECT(need(m, "missing.key", {1}, &p) == SVT_ERR_KEY && std::string(svt_last_error()).find("missing.key") != std::string::npos);
  EXPECT(find(m, "nope") == nullptr && find(m, "w") != nullptr);
  // ---- configuration validation: the good one passes, every field pushed out of range is refused (and nothing divides by zero)
  svt_encoder_config c = good_cfg();
  EXPECT(validate_cfg(c) == SVT_OK);
#define BAD(stmt) { svt_encoder_config b = good_cfg(); stmt; EXPECT(validate_cfg(b) == SVT_ERR_INVALID); }
  BAD(b.struct_size = 4) BAD(b.num_conv_layers = -1) BAD(b.num_conv_layers = SVT_MAX_CONV_LAYERS + 1) BAD(b.conv_kernel[0] = 9)
  BAD(b.conv_stride[0] = 6) BAD(b.conv_stride[0] = 0) BAD(b.conv_dim[3] = 12) BAD(b.conv_dim[3] = 0) BAD(b.conv_kernel[2] = 0)
  BAD(b.conv_stride[4] = -3) BAD(b.conv_dim[0] = 1024) BAD(b.num_heads = 0) BAD(b.num_heads = -12) BAD(b.num_heads = 7)
  BAD(b.hidden_size = 0) BAD(b.hidden_size = -768) BAD(b.pos_conv_groups = 0) BAD(b.pos_conv_groups = 5) BAD(b.intermediate_size = 3075)
  BAD(b.intermediate_size = 0) BAD(b.feat_extract_norm = 99) BAD(b.precision = 99) BAD(b.precision = -1) BAD(b.pos_conv_depth = 0)
  BAD(b.pos_conv_depth = 17) BAD(b.pos_conv_batch_norm = 1; b.pos_conv_depth = 2) BAD(b.rel_pos_buckets = 6) BAD(b.rel_pos_buckets = -4)
  BAD(b.rel_pos_buckets = 320; b.rel_pos_max_distance = 10) BAD(b.num_layers = -1) BAD(b.pos_conv_kernel = 0)
  BAD(b.num_conv_layers = 0; b.conv_dim[0] = 12) BAD(b.num_conv_layers = 0; b.conv_dim[0] = 128; b.normalize_wav = 1)
  BAD(b.hidden_size = 100; b.num_heads = 25)   // head_dim 4
#undef BAD
  { svt_encoder_config b = good_cfg(); b.num_conv_layers = 0; b.conv_dim[0] = 128; EXPECT(validate_cfg(b) == SVT_OK); }   // features-in mode
  // ---- frames -> notes: refusals and the edges of the scan
  std::vector<svt_frame> fr(5);
  for (auto& f : fr) { f.p_on = 0.f; f.p_off = 0.f; f.octave = 1; f.pitch_class = 2; }
  fr[1].p_on = 0.9f;
  double t_on[2], t_off[2];
  int32_t pitch[2], lo[2], hi[2];
  int64_t n[1] = {-7};
  EXPECT(svt_frames_to_notes(fr.data(), 1, 5, nullptr, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_OK && n[0] == 1);
  EXPECT(pitch[0] == 1 * 12 + 2 + 36 && lo[0] == 1 && hi[0] == 5 && t_on[0] == 0.02 && t_off[0] == 0.02 * 4);
  EXPECT(svt_frames_to_notes(fr.data(), 1, 5, nullptr, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 0, n) == SVT_ERR_INVALID);   // capacity 0
  const int64_t too_long[1] = {6}, negative[1] = {-1}, zero[1] = {0}, one[1] = {1};
  EXPECT(svt_frames_to_notes(fr.data(), 1, 5, too_long, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_ERR_INVALID);
  EXPECT(svt_frames_to_notes(fr.data(), 1, 5, negative, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_ERR_INVALID);
  EXPECT(svt_frames_to_notes(fr.data(), 1, 5, zero, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_OK && n[0] == 0);
  EXPECT(svt_frames_to_notes(fr.data() + 1, 1, 4, one, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_ERR_INVALID);   // the reference's empty window
  EXPECT(std::string(svt_last_error()).find("empty onset window") != std::string::npos);
  EXPECT(svt_frames_to_notes(nullptr, 1, 5, nullptr, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_ERR_INVALID);
  EXPECT(svt_frames_to_notes(fr.data(), -1, 5, nullptr, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_ERR_INVALID);
  EXPECT(svt_frames_to_notes(fr.data(), 1, 5, nullptr, 0.4f, 0.5f, 0.02, 0, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_ERR_INVALID);
  EXPECT(svt_frames_to_notes(fr.data(), 1, 5, nullptr, 0.4f, 0.5f, 0.02, 4, 100000, t_on, t_off, pitch, lo, hi, 2, n) == SVT_ERR_INVALID);
  fr[2].octave = 1000000; fr[2].pitch_class = 7;           // a class index outside the histogram
  EXPECT(svt_frames_to_notes(fr.data(), 1, 5, nullptr, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_ERR_INVALID);
  fr[2].octave = -5; fr[2].pitch_class = -9;
  EXPECT(svt_frames_to_notes(fr.data(), 1, 5, nullptr, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_ERR_INVALID);
  fr[2].octave = 2147483647; fr[2].pitch_class = 2147483647;   // octave * n_class must not overflow int
  EXPECT(svt_frames_to_notes(fr.data(), 1, 5, nullptr, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_ERR_INVALID);
  EXPECT(svt_frames_to_notes(fr.data(), 0, 5, nullptr, 0.4f, 0.5f, 0.02, 4, 12, t_on, t_off, pitch, lo, hi, 2, n) == SVT_OK);   // empty batch
  std::printf("hostile ok\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && !std::strcmp(argv[1], "notes")) return run_notes(argc, argv);
  if (argc >= 2 && !std::strcmp(argv[1], "hostile")) return run_hostile();
  std::fprintf(stderr, "usage: host_san_test notes <frames.bin> B T onset offset frame_size [capacity] | hostile\n");
  return 2;
}
