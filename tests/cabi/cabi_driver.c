/* A caller of libsvt_mi355.so that is NOT Python and has no torch: plain C, the HIP runtime for device memory, dlopen for
 * the library.  tests/test_gpu_cabi.py writes a blob (encoder config, parameters by HF key, a waveform batch), builds this
 * file with gcc and compares what it prints/writes with the Python binding on the same blob: the C-ABI declared in
 * include/svt_mi355.h is the whole boundary.
 *
 *   cabi_driver <libsvt_mi355.so> <blob.bin> <features_out.bin>
 *
 * blob: int32 magic 0x53565431 | svt_encoder_config (raw) | int32 n_params | n_params x { int32 key_len, key bytes,
 *       int32 ndim, int64 shape[ndim], float data[prod(shape)] } | int32 B | int64 L | float wav[B*L]
 * out : int32 B | int64 T | int32 D | float feats[B*T*D]
 */
#define __HIP_PLATFORM_AMD__ 1
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "svt_mi355.h"

#define DIE(...)                  \
  do {                            \
    fprintf(stderr, __VA_ARGS__); \
    fprintf(stderr, "\n");        \
    exit(1);                      \
  } while (0)
#define HIP_OK(x)                                                              \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) DIE("%s: %s", #x, hipGetErrorString(e_));            \
  } while (0)

typedef const char* (*last_error_fn)(void);
typedef int (*abi_version_fn)(void);
typedef int (*device_count_fn)(void);
typedef int (*create_fn)(const svt_encoder_config*, int, svt_encoder**);
typedef void (*destroy_fn)(svt_encoder*);
typedef int (*load_param_fn)(svt_encoder*, const char*, const void*, int, const int64_t*, int);
typedef int (*finalize_fn)(svt_encoder*);
typedef int64_t (*num_frames_fn)(const svt_encoder*, int64_t);
typedef int64_t (*workspace_fn)(const svt_encoder*, int32_t, int64_t);
typedef int (*forward_fn)(svt_encoder*, const float*, int32_t, int64_t, float*, void*, size_t, void*);

static void* sym(void* lib, const char* name) {
  void* p = dlsym(lib, name);
  if (!p) DIE("missing symbol %s", name);
  return p;
}

static void rd(FILE* f, void* p, size_t n) {
  if (fread(p, 1, n, f) != n) DIE("short read");
}

int main(int argc, char** argv) {
  if (argc != 4) DIE("usage: %s <lib.so> <blob.bin> <out.bin>", argv[0]);
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) DIE("dlopen: %s", dlerror());
  last_error_fn last_error = (last_error_fn)sym(lib, "svt_last_error");
  abi_version_fn abi_version = (abi_version_fn)sym(lib, "svt_abi_version");
  device_count_fn device_count = (device_count_fn)sym(lib, "svt_device_count");
  create_fn create = (create_fn)sym(lib, "svt_encoder_create");
  destroy_fn destroy = (destroy_fn)sym(lib, "svt_encoder_destroy");
  load_param_fn load_param = (load_param_fn)sym(lib, "svt_encoder_load_param");
  finalize_fn finalize = (finalize_fn)sym(lib, "svt_encoder_finalize");
  num_frames_fn num_frames = (num_frames_fn)sym(lib, "svt_encoder_num_frames");
  workspace_fn workspace_bytes = (workspace_fn)sym(lib, "svt_encoder_workspace_bytes");
  forward_fn forward = (forward_fn)sym(lib, "svt_encoder_forward");

  if (abi_version() != SVT_ABI_VERSION) DIE("ABI version %d, header says %d", abi_version(), SVT_ABI_VERSION);
  if (device_count() < 1) DIE("no gfx950 device");

  FILE* f = fopen(argv[2], "rb");
  if (!f) DIE("cannot open %s", argv[2]);
  int32_t magic;
  rd(f, &magic, 4);
  if (magic != 0x53565431) DIE("bad blob");
  svt_encoder_config cfg;
  rd(f, &cfg, sizeof cfg);
  if (cfg.struct_size != (int32_t)sizeof cfg) DIE("blob was written for another svt_encoder_config (%d vs %zu bytes)", cfg.struct_size, sizeof cfg);

  svt_encoder* enc = NULL;
  if (create(&cfg, 0, &enc) != SVT_OK) DIE("svt_encoder_create: %s", last_error());
  int32_t n_params;
  rd(f, &n_params, 4);
  for (int i = 0; i < n_params; ++i) {
    int32_t klen, ndim;
    char key[512];
    int64_t shape[8], numel = 1;
    rd(f, &klen, 4);
    if (klen <= 0 || klen >= (int)sizeof key) DIE("bad key length");
    rd(f, key, (size_t)klen);
    key[klen] = 0;
    rd(f, &ndim, 4);
    if (ndim < 0 || ndim > 8) DIE("bad ndim");
    rd(f, shape, 8u * (size_t)ndim);
    for (int d = 0; d < ndim; ++d) numel *= shape[d];
    float* data = (float*)malloc(sizeof(float) * (size_t)(numel > 0 ? numel : 1));
    rd(f, data, sizeof(float) * (size_t)numel);
    if (load_param(enc, key, data, SVT_F32, shape, ndim) != SVT_OK) DIE("svt_encoder_load_param(%s): %s", key, last_error());
    free(data);
  }
  if (finalize(enc) != SVT_OK) DIE("svt_encoder_finalize: %s", last_error());

  int32_t B;
  int64_t L;
  rd(f, &B, 4);
  rd(f, &L, 8);
  const size_t n_in = (size_t)B * (size_t)L;
  float* wav = (float*)malloc(sizeof(float) * n_in);
  rd(f, wav, sizeof(float) * n_in);
  fclose(f);

  const int64_t T = num_frames(enc, L);
  const int32_t D = cfg.hidden_size;
  const int64_t ws = workspace_bytes(enc, B, L);
  if (T < 1 || ws < 0) DIE("num_frames / workspace_bytes: %s", last_error());
  const size_t n_out = (size_t)B * (size_t)T * (size_t)D;

  HIP_OK(hipSetDevice(0));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  float *wav_dev = NULL, *feats_dev = NULL;
  void* ws_dev = NULL;
  HIP_OK(hipMalloc((void**)&wav_dev, sizeof(float) * n_in));
  HIP_OK(hipMalloc((void**)&feats_dev, sizeof(float) * n_out));
  HIP_OK(hipMalloc(&ws_dev, (size_t)ws));
  HIP_OK(hipMemcpyAsync(wav_dev, wav, sizeof(float) * n_in, hipMemcpyHostToDevice, stream));
  /* twice: the second call reuses the workspace (nothing is allocated or synchronised inside a forward) */
  for (int rep = 0; rep < 2; ++rep)
    if (forward(enc, wav_dev, B, L, feats_dev, ws_dev, (size_t)ws, (void*)stream) != SVT_OK) DIE("svt_encoder_forward: %s", last_error());
  float* feats = (float*)malloc(sizeof(float) * n_out);
  HIP_OK(hipMemcpyAsync(feats, feats_dev, sizeof(float) * n_out, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));

  /* error convention: a workspace that is too small is reported, not overrun */
  if (forward(enc, wav_dev, B, L, feats_dev, ws_dev, 16, (void*)stream) != SVT_ERR_WORKSPACE) DIE("expected SVT_ERR_WORKSPACE");

  FILE* o = fopen(argv[3], "wb");
  if (!o) DIE("cannot open %s", argv[3]);
  fwrite(&B, 4, 1, o);
  fwrite(&T, 8, 1, o);
  fwrite(&D, 4, 1, o);
  fwrite(feats, sizeof(float), n_out, o);
  fclose(o);
  double s = 0, s2 = 0;
  for (size_t i = 0; i < n_out; ++i) {
    s += feats[i];
    s2 += (double)feats[i] * feats[i];
  }
  printf("cabi_driver: B=%d T=%lld D=%d mean=%.6f meansq=%.6f\n", B, (long long)T, D, s / (double)n_out, s2 / (double)n_out);

  HIP_OK(hipFree(ws_dev));
  HIP_OK(hipFree(feats_dev));
  HIP_OK(hipFree(wav_dev));
  HIP_OK(hipStreamDestroy(stream));
  destroy(enc);
  free(feats);
  free(wav);
  dlclose(lib);
  return 0;
}
