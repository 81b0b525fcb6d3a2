#pragma once
#include "common.h"
#include "../../include/snerf_hip.h"

namespace snerf {

struct EncodeArgs {
  const float* rays = nullptr;   // (N,8) or null when xyz given
  const float* xyz = nullptr;    // (N,S,3) explicit positions
  const float* z = nullptr;      // (N,S)
  const float* sun_d = nullptr; int sun_stride = 3;
  const float* t = nullptr; const float* t_s = nullptr;
  int dir_is_sun = 0;            // solar-correction pass: x = o + sun_d * z
  int N = 0, S = 0, F = 0, Ep = 0;
  int FA = 0, W = 0, Xp = 0, x_sun = 0, x_t = 3, x_ts = -1, tau = 0;
  unsigned* zero = nullptr; int zero_n = 0;   // words the first workgroup clears (the pass's tile counters: one launch less than a kernel of their own)
};

struct CopyEntry {
  float* user; int user_ld; int rows; int cols; unsigned long long dst_off; int dst_ld;
};
constexpr int COPY_TABLE_MAX = 40;
struct CopyTable { CopyEntry e[COPY_TABLE_MAX]; int n; };

int launch_zero_bytes(void* p, size_t bytes, hipStream_t st);   // bytes % 4 == 0; a kernel, not a memset node (see aux_kernels.hip)
int launch_sample_z(const float* rays, const float* zsteps, const float* u, float* z, int N, int S, hipStream_t st);
int launch_copy_table(const CopyTable& tb, float* packed, int mode, hipStream_t st);
int launch_reduce_rows(const float* in, int n_in, size_t in_stride, int width, float* out, size_t out_stride,
                       int group, int accumulate, hipStream_t st);
int reduce_partials(const float* in, int n_in, size_t in_stride, int width, float* tmp, float* out, hipStream_t st);
// batched reductions (aux_kernels.hip): out[e] += sum_q in[q * stride + e], all jobs of a pass in two launches
// (cols / in_ld / out_ld: a 2-D job -- element e = r * cols + c lies at in[q * stride + r * in_ld + c] and goes to out[r * out_ld + c];
//  1-D jobs have cols = width)
struct RedJob { const float* in; float* out; unsigned long long stride; long long width; int n_in; int blk0; int vec; int cols; int in_ld; int out_ld; int pad[2]; };
constexpr int RED_MAX = 56;
struct RedTable { RedJob j[RED_MAX]; int n = 0; int blocks = 0; };      // host-side queue
constexpr int RED_CHUNK = 24;
struct RedChunk { RedJob j[RED_CHUNK]; int n = 0; };                      // what one launch carries as its argument
int red_add_elem(RedTable& tb, const float* in, int n_in, size_t stride, size_t width, float* out);   // few slabs, many elements
int red_add_elem2d(RedTable& tb, const float* in, int n_in, size_t stride, int rows, int cols, int in_ld, float* out, int out_ld);   // a column block of a wider matrix
int red_add_col(RedTable& tb, const float* in, int n_in, size_t stride, int width, float* out);       // many partial rows, <= ~1024 columns
int launch_reductions(const RedTable& elem, const RedTable& col, hipStream_t st);
int launch_ray_sum32(const float* d32, int col0, int N, int S, int tau, float* out, hipStream_t st);   // out[n][c] = sum_s d32[(n S + s)][col0 + c], rows of 32 floats
int launch_embedding_rows(const float* table, int n_embed, int tau, const long long* idx, int n, float* rows, hipStream_t st);
int launch_embedding_backward(const long long* idx, const float* d_rows, int n, int tau, int n_embed, float* grad, hipStream_t st);

}  // namespace snerf
