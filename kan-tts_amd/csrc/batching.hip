// Batch assembly on the device: padding and random cropping of ragged utterances that are RESIDENT in
// HBM.  A 288 GB device holds a whole TTS corpus (10 h of 22.05 kHz audio + its mels is ~4 GB), so the per-step host work
// of the reference's collate functions -- np.pad per utterance, np.stack, torch.tensor(...), a pinned copy, H2D
// (kantts/datasets/dataset.py:34-85 Padder, :278-311 Voc_Dataset.collate_fn, :690-827 AM_Dataset.collate_fn) -- shrinks to
// uploading B row offsets / crop starts / lengths; the gather, the padding and the (frames, C) -> (C, frames) transpose of
// the vocoder's mel crop happen here:
//     out[b][t][c] = t < len[b] ? src[(row_off[b] + start[b] + t) * C + c] : pad[c]        (transpose: out[b][c][t])
// One thread per output element; a training batch is ~1-5 MB, the kernel is a few microseconds.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void ragged_rows_kernel(const T* __restrict__ src, const long long* __restrict__ row_off,
                                                         const int* __restrict__ start, const int* __restrict__ len,
                                                         const T* __restrict__ pad, T* __restrict__ out, int Tmax, int C,
                                                         int transpose) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)Tmax * C) return;
  int t, c;
  if (transpose) {
    c = (int)(e / Tmax);
    t = (int)(e - (long long)c * Tmax);
  } else {
    t = (int)(e / C);
    c = (int)(e - (long long)t * C);
  }
  T v = pad ? pad[c] : (T)0;
  if (t < len[b]) v = src[(row_off[b] + (start ? start[b] : 0) + t) * C + c];
  out[(long long)b * Tmax * C + e] = v;
}

template <typename T>
static int rr_launch(const T* src, const long long* row_off, const int* start, const int* len, const T* pad, T* out, int B,
                     int Tmax, int C, int transpose, void* stream) {
  if (!src || !row_off || !len || !out || B < 0 || Tmax < 0 || C < 1) return KANTTS_E_BADARG;
  if (B == 0 || Tmax == 0) return KANTTS_OK;
  if (B > 65535) return KANTTS_E_UNSUPPORTED;
  dim3 grid((unsigned)kantts_cdiv((long long)Tmax * C, 256), B);
  hipLaunchKernelGGL((ragged_rows_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, src, row_off, start, len, pad, out, Tmax, C,
                     transpose);
  KANTTS_CHECK_LAUNCH();
}

extern "C" int kantts_ragged_rows_f32(const float* src, const int64_t* row_off, const int32_t* start, const int32_t* len,
                                      const float* pad, float* out, int B, int Tmax, int C, int transpose, void* stream) {
  return rr_launch<float>(src, reinterpret_cast<const long long*>(row_off), start, len, pad, out, B, Tmax, C, transpose, stream);
}

extern "C" int kantts_ragged_rows_i64(const int64_t* src, const int64_t* row_off, const int32_t* start, const int32_t* len,
                                      const int64_t* pad, int64_t* out, int B, int Tmax, int C, int transpose, void* stream) {
  return rr_launch<long long>(reinterpret_cast<const long long*>(src), reinterpret_cast<const long long*>(row_off), start, len,
                              reinterpret_cast<const long long*>(pad), reinterpret_cast<long long*>(out), B, Tmax, C,
                              transpose, stream);
}
