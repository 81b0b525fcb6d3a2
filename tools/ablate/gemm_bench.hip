// Diagnostic harness (not part of the product): times launch_gemm on the headline layer shapes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../../semantic-nerf-for-satellite-data_amd/csrc/gemm.h"
namespace snerf { void set_error(const char* fmt, ...) {} }
using namespace snerf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static double time_gemm(const GemmArgs& g, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch_gemm(g, 0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) launch_gemm(g, 0);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const bool X6 = argc > 1 && argv[1][0] == 'x';
  const int PLANES = argc > 2 ? atoi(argv[2]) : 3;
  const int TILE = argc > 3 ? atoi(argv[3]) : 0;
  (void)(argc > 4 ? atoi(argv[4]) : 0);   // (former operand-format switch: the fp16-plane arithmetic now lives in bsp_gemm.hip, see build_one.sh)
  const int P = 262144, W = 512;
  float *X, *Wt, *Y, *Y2, *dW, *cs;
  CK(hipMalloc(&X, (size_t)P * W * 4)); CK(hipMalloc(&Wt, (size_t)W * W * 4)); CK(hipMalloc(&Y, (size_t)P * W * 4));
  CK(hipMalloc(&Y2, (size_t)P * W * 4)); CK(hipMalloc(&dW, (size_t)128 * W * W * 4)); CK(hipMalloc(&cs, (size_t)(P / 32) * W * 4));
  std::vector<float> h((size_t)P * W);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Y2, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Wt, h.data(), (size_t)W * W * 4, hipMemcpyHostToDevice));
  unsigned short* Bplanes; CK(hipMalloc(&Bplanes, (size_t)3 * W * W * 2 + 4096)); CK(hipMemset(Bplanes, 0, (size_t)3 * W * W * 2));
  const double fl = 2.0 * P * W * W;
  if (argc > 6 && atoi(argv[6]) >= 1) {   // counter runs: ONE kernel configuration only (1: fwd plain, 2: dW split 64)
    GemmArgs g;
    if (atoi(argv[6]) == 1) { g.A = X; g.lda = W; g.B = Wt; g.ldb = W; g.I = P; g.J = W; g.K = W; g.C = Y; g.ldc = W;
      if (X6) { g.Bpl = Bplanes; g.pl_stride = (size_t)W * W; g.bt_rows = W; g.bt_elems = (size_t)W * W; } }
    else { g.A = X; g.lda = W; g.a_ic = true; g.B = Y2; g.ldb = W; g.b_ic = true; g.I = W; g.J = W; g.K = P; g.C = dW; g.ldc = W;
      g.k_split = ((P + 63) / 64 + 31) / 32 * 32; g.n_split = (P + g.k_split - 1) / g.k_split; g.slab_stride = (size_t)W * W; }
    g.x6 = X6; g.planes = PLANES; g.tile = TILE;
    double t = time_gemm(g, 12); printf("only-mode %d: %.3f ms\n", atoi(argv[6]), t);
    return 0;
  }
  { GemmArgs g; g.A = X; g.lda = W; g.B = Wt; g.ldb = W; g.I = P; g.J = W; g.K = W; g.C = Y; g.ldc = W; g.x6 = X6; g.planes = PLANES; g.tile = TILE;
    time_gemm(g, 400); }  // clock / power state warm-up
  { GemmArgs g; g.A = X; g.lda = W; g.B = Wt; g.ldb = W; g.I = P; g.J = W; g.K = W; g.C = Y; g.ldc = W; g.x6 = X6; g.planes = PLANES; g.tile = TILE;
    if (X6) { g.Bpl = Bplanes; g.pl_stride = (size_t)W * W; g.bt_rows = W; g.bt_elems = (size_t)W * W; }
    double t = time_gemm(g, 20); printf("fwd plain        %.3f ms %.1f TF\n", t, fl / t / 1e9);
    g.bias = Wt; g.act = ACT_SIN; g.w0 = 1.f;
    t = time_gemm(g, 20); printf("fwd sin          %.3f ms %.1f TF\n", t, fl / t / 1e9);
    g.C2 = Y2;
    t = time_gemm(g, 20); printf("fwd sin+cos out  %.3f ms %.1f TF\n", t, fl / t / 1e9); }
  { GemmArgs g; g.A = X; g.lda = W; g.B = Wt; g.ldb = W; g.b_ic = true; g.I = P; g.J = W; g.K = W; g.C = Y; g.ldc = W; g.x6 = X6; g.planes = PLANES; g.tile = TILE;
    double t = time_gemm(g, 20); printf("dX plain         %.3f ms %.1f TF\n", t, fl / t / 1e9);
    g.aux = Y2; g.ldaux = W; g.aux_mode = AUX_MUL; g.colsum = cs; g.ldcs = W;
    t = time_gemm(g, 20); printf("dX aux+colsum    %.3f ms %.1f TF\n", t, fl / t / 1e9); }
  for (int ns : {32, 48, 64, 128}) { GemmArgs g; g.A = X; g.lda = W; g.a_ic = true; g.B = Y2; g.ldb = W; g.b_ic = true; g.I = W; g.J = W; g.K = P; g.C = dW; g.ldc = W; g.x6 = X6; g.planes = PLANES; g.tile = TILE;
    g.k_split = ((P + ns - 1) / ns + 31) / 32 * 32; g.n_split = (P + g.k_split - 1) / g.k_split; g.slab_stride = (size_t)W * W;
    double t = time_gemm(g, 20); printf("dW split %d      %.3f ms %.1f TF\n", ns, t, fl / t / 1e9); }
  for (int div : {1, 2, 4, 8, 16}) {  // size sweep: fixed per-launch cost?
    GemmArgs g; g.A = X; g.lda = W; g.B = Wt; g.ldb = W; g.I = P / div; g.J = W; g.K = W; g.C = Y; g.ldc = W; g.x6 = X6; g.planes = PLANES; g.tile = TILE;
    if (X6) { g.Bpl = Bplanes; g.pl_stride = (size_t)W * W; g.bt_rows = W; g.bt_elems = (size_t)W * W; }
    double t = time_gemm(g, 40); printf("fwd plain I=P/%-2d  %.3f ms %.1f TF\n", div, t, fl / div / t / 1e9);
  }
#ifdef SNERF_ABL_CLOCK
  {  // per-workgroup phase cycles of the split kernel (colsum doubles as the stamp buffer; the epilogue skips colsum under CLOCK)
    GemmArgs g; g.A = X; g.lda = W; g.B = Wt; g.ldb = W; g.I = P; g.J = W; g.K = W; g.C = Y; g.ldc = W; g.x6 = X6; g.planes = PLANES; g.tile = TILE;
    if (X6) { g.Bpl = Bplanes; g.pl_stride = (size_t)W * W; g.bt_rows = W; g.bt_elems = (size_t)W * W; }
    if (argc <= 5 || atoi(argv[5]) != 0) { g.bias = Wt; g.act = ACT_SIN; g.w0 = 1.f; }
    unsigned long long* dbg; const int nb = (P / 128) * (W / 128);
    CK(hipMalloc(&dbg, (size_t)nb * 64)); CK(hipMemset(dbg, 0, (size_t)nb * 64));
    time_gemm(g, 300);
    g.colsum = (float*)dbg; g.ldcs = 4;
    double t = time_gemm(g, 20);
    std::vector<unsigned long long> hs((size_t)nb * 8);
    CK(hipMemcpy(hs.data(), dbg, hs.size() * 8, hipMemcpyDeviceToHost));
    auto med = [&](int f) { std::vector<double> v; for (int i = 0; i < nb; ++i) if (hs[8 * i + 6] > 0) v.push_back((double)hs[8 * i + f]); std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    printf("fwd sin %.3f ms; median cycles per workgroup: prologue %.0f  k-loop %.0f  epilogue issue %.0f  store drain (vmcnt 0) %.0f  total %.0f\n", t, med(0), med(1), med(2), med(3), med(6));
    unsigned long long r0 = ~0ull, r1 = 0; double sum = 0;
    for (int i = 0; i < nb; ++i) if (hs[8 * i + 6] > 0) { r0 = std::min(r0, hs[8 * i + 4]); r1 = std::max(r1, hs[8 * i + 5]); sum += (double)(hs[8 * i + 5] - hs[8 * i + 4]); }
    printf("launch span %.1f us; mean resident workgroups per CU %.2f; clock %.2f GHz\n", (r1 - r0) * 0.01, sum / (double)(r1 - r0) / 256.0, med(6) / ((sum / nb) * 10.0));
  }
#endif
  return 0;
}
