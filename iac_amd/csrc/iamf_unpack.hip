// iamf_unpack.hip — LPCM sub-stream packets -> planar f32 element PCM on the device (include/iamf_hip.h:
// iamf_hip_lpcm_unpack).  The role of the reference's LPCM "decoder" (src/iamf_dec/pcm/IAMF_pcm_decoder.c:64-83,
// 133-149: one sample = integer / 2^(bits-1), 16 / 24 / 32 bit, little- or big-endian with the reference's own
// big-endian 24-bit byte order, bitstream.c:204-208) and of the channel re-ordering behind it (audio-layer order ->
// the renderer's order, IAMF_decoder.c:2230-2260), for the group of decoder handles: the host uploads the packets as
// they are in the stream, half (16 bit) of the bytes of unpacked f32, and spends no time converting.
// Integer -> float conversions are exact or round to nearest exactly as the CPU's; the scale is a power of two.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/iamf_hip.h"

namespace {

struct UnpackParams {
  iamf_hip_lpcm_layout lay;
  const uint8_t *raw;
  int64_t raw_stride;
  const int32_t *first_count;   // per stream {first sample (a trimmed start), samples to write}, fc_stride int32 apart
  int64_t fc_stride;
  float *out;
  int64_t out_stride;
  int32_t n_streams;
  int32_t n_frames;             // frames per stream in one launch (iamf_hip_batch_render_lpcm); the public entry: 1
  int64_t raw_frame_stride;     // bytes from one frame's packet row to the next
  int64_t out_frame_stride;     // floats
  int32_t fc_uniform[2];        // first_count == nullptr: this pair for every stream, carried in the kernel arguments
};

template <int BYTES>
__device__ __forceinline__ float lpcm_value(const uint8_t *p, bool le) {
  if (BYTES == 2) {
    const int v = le ? (int)(int16_t)(p[0] | (p[1] << 8)) : (int)(int16_t)(p[1] | (p[0] << 8));
    return (float)v * (1.0f / 32768.0f);
  } else if (BYTES == 3) {
    int v = le ? (p[0] | (p[1] << 8) | (p[2] << 16)) : (p[2] | (p[0] << 8) | (p[1] << 16));   // reads24be: byte 1 is the top one
    if (v & 0x800000) v |= ~0xffffff;
    return (float)v * (1.0f / 8388608.0f);
  } else {
    const uint32_t u = le ? ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24))
                          : ((uint32_t)p[3] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 24));
    return (float)(int32_t)u * (1.0f / 2147483648.0f);   // int -> float rounds to nearest even, as the CPU's cvtsi2ss
  }
}

// grid (quads of samples / 256, channels, streams x frames); a thread = four consecutive samples of one channel
// (256 threads: a 1024-sample frame is one workgroup per channel — with 64 the launch was bound by the rate at which
//  workgroups are dispatched: 4 M workgroups for 1024 streams x 64 frames x 16 channels)
template <int BYTES>
__global__ __launch_bounds__(256) void lpcm_unpack_kernel(const UnpackParams p) {
  const int s = blockIdx.z / p.n_frames, fr = blockIdx.z - s * p.n_frames, c = blockIdx.y;
  const int i0 = 4 * (blockIdx.x * 256 + threadIdx.x);
  // {first, count} come from the host per call; whatever they hold, no thread reads outside the frame's packet
  // (the bytes of samples [0, frame_size) are what iamf_hip_lpcm_unpack checked against the raw stride)
  int first = p.first_count ? p.first_count[s * p.fc_stride] : p.fc_uniform[0];
  int count = p.first_count ? p.first_count[s * p.fc_stride + 1] : p.fc_uniform[1];
  first = first < 0 ? 0 : (first > p.lay.frame_size ? p.lay.frame_size : first);
  count = count < p.lay.frame_size - first ? count : p.lay.frame_size - first;
  if (i0 >= count) return;
  const int off = p.lay.src_offset[c], step = p.lay.src_step[c];
  float *dst = p.out + (int64_t)s * p.out_stride + (int64_t)fr * p.out_frame_stride + (int64_t)c * p.lay.frame_size + i0;
  const uint8_t *src = p.raw + (int64_t)s * p.raw_stride + (int64_t)fr * p.raw_frame_stride + off + (int64_t)(first + i0) * step;
  const bool le = p.lay.little_endian != 0;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (off >= 0 && i0 + j < count) ? lpcm_value<BYTES>(src + (int64_t)j * step, le) : 0.f;
  if (i0 + 4 <= count) {
    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);   // frame_size and i0 are multiples of 4
  } else {
    for (int j = 0; i0 + j < count; ++j) dst[j] = v[j];
  }
}

// 16 bytes per lane, one pass: what hipMemcpyAsync would move, without leaving the compute queue
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__global__ __launch_bounds__(256) void upload_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = __builtin_nontemporal_load(src + i);
}

// one lane, one word of pinned host memory behind a system-scope fence: "everything queued on this stream before me is done"
__global__ void signal_kernel(volatile uint32_t *flag, uint32_t seq) {
  __threadfence_system();
  *flag = seq;
}

}  // namespace

extern "C" int iamf_hip_stream_signal(void *stream, volatile uint32_t *h_pinned_flag, uint32_t seq) {
  if (!h_pinned_flag) return IAMF_HIP_ERR_BAD_ARG;
  hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), h_pinned_flag, seq);
  return hipGetLastError() == hipSuccess ? IAMF_HIP_OK : IAMF_HIP_ERR_DEVICE;
}

extern "C" int iamf_hip_upload_by_kernel(const void *h_pinned, void *d_dst, size_t bytes, void *stream) {
  if (!h_pinned || !d_dst || !bytes || (bytes & 15) || ((uintptr_t)h_pinned & 15) || ((uintptr_t)d_dst & 15)) return IAMF_HIP_ERR_BAD_ARG;
  const size_t n16 = bytes / 16;
  if ((n16 + 255) / 256 > 0x7fffffffu) return IAMF_HIP_ERR_BAD_ARG;
  hipLaunchKernelGGL(upload_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const u32x4 *>(h_pinned), static_cast<u32x4 *>(d_dst), n16);
  return hipGetLastError() == hipSuccess ? IAMF_HIP_OK : IAMF_HIP_ERR_DEVICE;
}

// n_frames frames per stream in one launch: frame f of stream s reads its packet row at d_raw + s * raw_stream_stride +
// f * raw_frame_stride and writes d_out + s * out_stream_stride + f * out_frame_stride (iamf_hip_batch_render_lpcm's
// general form; one {first, count} pair serves all frames of a stream).  The public entry below is n_frames = 1.
// d_first_count == nullptr: {uniform_first, uniform_count} for every stream, passed with the launch (no device word to keep
// coherent with launches still queued: iamf_hip_batch_render_lpcm_range's unfused form).
extern "C" __attribute__((visibility("hidden"))) int iamf_hip_lpcm_unpack_frames(
    const iamf_hip_lpcm_layout *lay, const void *d_raw, int64_t raw_stream_stride, int64_t raw_frame_stride, int32_t n_frames,
    const int32_t *d_first_count, int64_t first_count_stride, float *d_out, int64_t out_stream_stride, int64_t out_frame_stride,
    int32_t n_streams, void *stream, int32_t uniform_first, int32_t uniform_count) {
  if (!lay || !d_raw || (d_first_count && first_count_stride < 2 && first_count_stride != 0) || !d_out || n_streams <= 0 ||
      n_frames <= 0 || (int64_t)n_streams * n_frames > 65535)
    return IAMF_HIP_ERR_BAD_ARG;
  if (lay->sample_bytes < 2 || lay->sample_bytes > 4 || lay->channels <= 0 || lay->channels > IAMF_HIP_LPCM_MAX_CHANNELS ||
      lay->frame_size <= 0 || (lay->frame_size & 3) || raw_frame_stride <= 0 || raw_stream_stride < (int64_t)n_frames * raw_frame_stride ||
      out_frame_stride < (int64_t)lay->channels * lay->frame_size || out_stream_stride < (int64_t)n_frames * out_frame_stride)
    return IAMF_HIP_ERR_BAD_ARG;
  for (int c = 0; c < lay->channels; ++c) {
    // every byte a thread may read lies inside the frame's packet row (first + count <= frame_size: clamped by the kernel)
    if (lay->src_offset[c] < 0) continue;
    if (lay->src_step[c] < lay->sample_bytes ||
        (int64_t)lay->src_offset[c] + (int64_t)(lay->frame_size - 1) * lay->src_step[c] + lay->sample_bytes > raw_frame_stride)
      return IAMF_HIP_ERR_BAD_ARG;
  }
  UnpackParams p;
  p.lay = *lay;
  p.raw = static_cast<const uint8_t *>(d_raw);
  p.raw_stride = raw_stream_stride;
  p.first_count = d_first_count;
  p.fc_stride = first_count_stride;
  p.out = d_out;
  p.out_stride = out_stream_stride;
  p.n_streams = n_streams;
  p.n_frames = n_frames;
  p.raw_frame_stride = raw_frame_stride;
  p.out_frame_stride = out_frame_stride;
  p.fc_uniform[0] = uniform_first;
  p.fc_uniform[1] = uniform_count;
  const dim3 grid((unsigned)((lay->frame_size / 4 + 255) / 256), (unsigned)lay->channels, (unsigned)(n_streams * n_frames));
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (lay->sample_bytes) {
    case 2: hipLaunchKernelGGL(lpcm_unpack_kernel<2>, grid, dim3(256), 0, st, p); break;
    case 3: hipLaunchKernelGGL(lpcm_unpack_kernel<3>, grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL(lpcm_unpack_kernel<4>, grid, dim3(256), 0, st, p); break;
  }
  return hipGetLastError() == hipSuccess ? IAMF_HIP_OK : IAMF_HIP_ERR_DEVICE;
}

extern "C" int iamf_hip_lpcm_unpack(const iamf_hip_lpcm_layout *lay, const void *d_raw, int64_t raw_stream_stride,
                                    const int32_t *d_first_count, int64_t first_count_stride, float *d_out,
                                    int64_t out_stream_stride, int32_t n_streams, void *stream) {
  // one frame per stream: the stream's raw region is the frame's packet row
  if (!d_first_count) return IAMF_HIP_ERR_BAD_ARG;
  return iamf_hip_lpcm_unpack_frames(lay, d_raw, raw_stream_stride, raw_stream_stride, 1, d_first_count, first_count_stride, d_out,
                                     out_stream_stride, out_stream_stride, n_streams, stream, 0, 0);
}

// Sample-frames [stream][sample][channels] (what a batch with out_format F32 writes) -> planar [stream][channel][...]
// (what a batch reads as element PCM).  A workgroup takes 256 sample-frames of one stream through LDS: contiguous reads,
// contiguous writes per channel; the LDS row stride is odd, so the transposed reads spread over the banks.
__global__ __launch_bounds__(256) void deinterleave_kernel(const float *src, int64_t src_stream_stride, int channels, int n,
                                                           float *dst, int64_t dst_stream_stride, int64_t dst_channel_stride) {
  __shared__ float tile[256 * 25];
  const int s = blockIdx.y, j0 = 256 * (int)blockIdx.x, t = threadIdx.x;
  const int rows = n - j0 < 256 ? n - j0 : 256, ld = channels | 1;
  const float *in = src + (int64_t)s * src_stream_stride + (int64_t)j0 * channels;
  for (int k = t; k < rows * channels; k += 256) tile[(k / channels) * ld + k % channels] = in[k];
  __syncthreads();
  float *out = dst + (int64_t)s * dst_stream_stride + j0;
  if (t < rows)
    for (int c = 0; c < channels; ++c) out[(int64_t)c * dst_channel_stride + t] = tile[t * ld + c];
}

extern "C" int iamf_hip_deinterleave_f32(const float *d_src, int64_t src_stream_stride, int32_t channels, int32_t n_streams,
                                         int32_t n_samples, float *d_dst, int64_t dst_stream_stride, int64_t dst_channel_stride,
                                         void *stream) {
  if (!d_src || !d_dst || channels <= 0 || channels > 24 || n_streams <= 0 || n_streams > 65535 || n_samples < 0 ||
      dst_channel_stride < n_samples || (n_streams > 1 && (src_stream_stride < (int64_t)n_samples * channels ||
                                                           dst_stream_stride < (int64_t)channels * dst_channel_stride)))
    return IAMF_HIP_ERR_BAD_ARG;
  if (!n_samples) return IAMF_HIP_OK;
  hipLaunchKernelGGL(deinterleave_kernel, dim3((unsigned)((n_samples + 255) / 256), (unsigned)n_streams), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d_src, src_stream_stride, channels, n_samples, d_dst, dst_stream_stride,
                     dst_channel_stride);
  return hipGetLastError() == hipSuccess ? IAMF_HIP_OK : IAMF_HIP_ERR_DEVICE;
}
