// chain3_body (pygradflow_amd/csrc/pgf_chain3.h) alone: LDL^T of one diagonal block against a
// host factorisation -- L, D, inertia, the 64 x 64 tile inverses -- for full and ragged block
// sizes, with the kernel's time and its per-step stamps.
//   hipcc -O3 --offload-arch=gfx950 -I pygradflow_amd/csrc -o tools/bin/chain3_test tools/chain3_test.hip
#include "pgf_chain3.h"

#include <cmath>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void k_chain3(double *K, int64_t ldk, int c0, int nb, double *dvec,
                                                 double *dinv, int *flags, double *Linv, double *LinvT,
                                                 long long *dbg) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[C3_SMEM];
  chain3_body(smem, K, ldk, c0, nb, dvec, dinv, flags, Linv, LinvT, dbg);
}

// touches the block the way the previous launch of the factorisation leaves it: in L2
__global__ void k_touch(double *K, int64_t ldk, int c0, int nb) {
  for (int p = threadIdx.x + blockIdx.x * blockDim.x; p < nb * nb; p += blockDim.x * gridDim.x) {
    const int i = p / nb, j = p % nb;
    if (j <= i) K[(int64_t)(c0 + i) * ldk + c0 + j] += 0.0;
  }
}

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main() {
  const int sizes[] = {256, 256, 240, 200, 130, 64, 17, 16, 5};
  const int c0 = 256;  // block offset inside a larger matrix
  const int NT = 576, ldk = 592;
  int fails = 0;
  for (int t = 0; t < (int)(sizeof(sizes) / sizeof(int)); ++t) {
    const int nb = sizes[t];
    std::vector<double> A((size_t)NT * ldk, 0.0);
    // symmetric, quasi-definite flavour: positive diagonal for the first 3/4 of the block,
    // negative for the rest; everything outside the block is poison
    unsigned s = 12345u + 77u * t;
    auto rnd = [&]() {
      s = s * 1664525u + 1013904223u;
      return ((s >> 8) & 0xffff) / 65536.0 - 0.5;
    };
    for (auto &v : A) v = 1e300;
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j <= i; ++j) {
        double v = 0.4 * rnd();
        if (i == j) v = (i < 3 * nb / 4) ? 6.0 + rnd() : -(5.0 + rnd());
        A[(size_t)(c0 + i) * ldk + c0 + j] = v;
      }
    // host LDL^T (lower, in place on a copy)
    std::vector<double> M((size_t)nb * nb, 0.0), L((size_t)nb * nb, 0.0), D(nb);
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j <= i; ++j) M[(size_t)i * nb + j] = A[(size_t)(c0 + i) * ldk + c0 + j];
    int neg_ref = 0;
    for (int c = 0; c < nb; ++c) {
      D[c] = M[(size_t)c * nb + c];
      neg_ref += D[c] < 0.0;
      for (int i = c + 1; i < nb; ++i) L[(size_t)i * nb + c] = M[(size_t)i * nb + c] / D[c];
      for (int i = c + 1; i < nb; ++i)
        for (int j = c + 1; j <= i; ++j) M[(size_t)i * nb + j] -= L[(size_t)i * nb + c] * M[(size_t)j * nb + c];
    }
    double *dK, *dvec, *dinv, *Linv, *LinvT;
    int *flags;
    long long *dbg;
    const int ntile = NT / 64;
    CK(hipMalloc(&dK, A.size() * 8));
    CK(hipMalloc(&dvec, NT * 8));
    CK(hipMalloc(&dinv, NT * 8));
    CK(hipMalloc(&Linv, (size_t)ntile * 4096 * 8));
    CK(hipMalloc(&LinvT, (size_t)ntile * 4096 * 8));
    CK(hipMalloc(&flags, 16));
    CK(hipMalloc(&dbg, 128 * 8));
    float best = 1e30f;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemcpy(dK, A.data(), A.size() * 8, hipMemcpyHostToDevice));
      CK(hipMemset(flags, 0, 16));
      CK(hipMemset(dbg, 0, 128 * 8));
      CK(hipMemset(Linv, 0xff, (size_t)ntile * 4096 * 8));
      CK(hipMemset(LinvT, 0xff, (size_t)ntile * 4096 * 8));
      hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, 0, dK, (int64_t)ldk, c0, nb);
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_chain3, dim3(1), dim3(1024), 0, 0, dK, (int64_t)ldk, c0, nb, dvec, dinv, flags,
                         Linv, LinvT, dbg);
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    std::vector<double> R(A.size()), hd(NT), hi(NT), hLi((size_t)ntile * 4096), hLiT((size_t)ntile * 4096);
    int hf[4];
    long long hs[128];
    CK(hipMemcpy(R.data(), dK, R.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hd.data(), dvec, NT * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hi.data(), dinv, NT * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hLi.data(), Linv, hLi.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hLiT.data(), LinvT, hLiT.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hf, flags, 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs, dbg, sizeof(hs), hipMemcpyDeviceToHost));
    double eL = 0.0, eD = 0.0, eI = 0.0, eInv = 0.0, eT = 0.0, eOut = 0.0;
    for (int i = 0; i < nb; ++i) {
      eD = std::max(eD, std::fabs(hd[c0 + i] - D[i]) / std::fabs(D[i]));
      eD = std::max(eD, std::fabs(R[(size_t)(c0 + i) * ldk + c0 + i] - D[i]) / std::fabs(D[i]));
      eI = std::max(eI, std::fabs(hi[c0 + i] * D[i] - 1.0));
      for (int j = 0; j < i; ++j)
        eL = std::max(eL, std::fabs(R[(size_t)(c0 + i) * ldk + c0 + j] - L[(size_t)i * nb + j]));
    }
    // nothing outside the block's lower triangle may change
    for (int i = 0; i < NT; ++i)
      for (int j = 0; j < ldk; ++j) {
        const bool inside = i >= c0 && i < c0 + nb && j >= c0 && j <= i;
        if (!inside && R[(size_t)i * ldk + j] != A[(size_t)i * ldk + j]) eOut += 1.0;
      }
    // tile inverses: Lt * inv = I for every 64 x 64 diagonal tile (identity-padded)
    const int ngr = (nb + 63) / 64;
    for (int g = 0; g < ngr; ++g) {
      const double *inv = hLi.data() + (size_t)(c0 / 64 + g) * 4096;
      const double *invT = hLiT.data() + (size_t)(c0 / 64 + g) * 4096;
      for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) {
          double sum = 0.0;
          for (int k = 0; k < 64; ++k) {
            const int gi = 64 * g + i, gk = 64 * g + k;
            double l = (i == k) ? 1.0 : 0.0;
            if (gi < nb && gk < nb && k < i) l = L[(size_t)gi * nb + gk];
            sum += l * inv[k * 64 + j];
          }
          eInv = std::max(eInv, std::fabs(sum - (i == j ? 1.0 : 0.0)));
          eT = std::max(eT, std::fabs(inv[i * 64 + j] - invT[j * 64 + i]));
        }
    }
    const bool ok = eL < 1e-12 && eD < 1e-12 && eI < 1e-14 && eInv < 1e-11 && eT == 0.0 && eOut == 0.0 &&
                    hf[0] == 0 && hf[1] == neg_ref;
    fails += !ok;
    printf("nb=%3d  %6.1f us  errL %.1e errD %.1e err1/D %.1e inv %.1e invT %.1e outside %g  flags %d neg %d (ref %d)  %s\n",
           nb, best * 1e3, eL, eD, eI, eInv, eT, eOut, hf[0], hf[1], neg_ref, ok ? "ok" : "FAIL");
    if (t == 0) {
      printf("  stamps (us since start):");
      for (int q = 1; q < 32 && hs[q]; ++q) printf(" %.2f", (hs[q] - hs[0]) * 0.01);
      printf("\n  shader cycles between stamps:");
      for (int q = 2; q < 32 && hs[q]; ++q) printf(" %lld", hs[64 + q] - hs[64 + q - 1]);
      printf("\n  owner wavefront 5 (steps 1 and 12: start, pass 2 done, B2, panel done, B3, column k+1 done, B1):");
      for (int q = 32; q < 64 && hs[q]; ++q) printf(" %.2f", (hs[q] - hs[0]) * 0.01);
      printf("\n");
    }
    hipFree(dK); hipFree(dvec); hipFree(dinv); hipFree(Linv); hipFree(LinvT); hipFree(flags); hipFree(dbg);
  }
  printf(fails ? "FAILED\n" : "all ok\n");
  return fails != 0;
}
