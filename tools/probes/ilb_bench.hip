// ilb_bench.hip -- stand-alone timing harness of ilb_kernel (sod100k_amd/csrc/k_ilb.hip) on synthetic data: launch time per
// ILBlock geometry of csnet-L-x2 at batch 64 and (with -DILB_TIMING) where a block spends its life, phase by phase.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DILB_TIMING -I sod100k_amd/csrc -o tools/probes/ilb_bench tools/probes/ilb_bench.hip
#include "../../sod100k_amd/csrc/k_ilb.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

static float* dalloc(size_t n, float v) {
  float* p; (void)hipMalloc(&p, n * 4);
  std::vector<float> h(n);
  unsigned s = 12345u + (unsigned)n;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = v * ((float)(s >> 8) / 8388608.f - 1.f); }
  (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
  return p;
}

static void run(const char* name, int B, int CH, int CL, int OH, int OL, int Hl, int nt, int k3 = 0, int ntl = -1) {
  IlbArgs a = {};
  a.CH = CH; a.CL = k3 ? CH : CL; a.OH = OH; a.OL = OL; a.Hl = Hl; a.Wl = Hl; a.B = B; a.Rh = 4; a.Rl = 4;
  a.k3 = k3;
  a.nth = nt; a.ntl = ntl >= 0 ? ntl : (OL > 0 ? nt : 0);
  const size_t lds = csn_ilb_layout(a);
  if (lds == 0 || lds > 160 * 1024) { printf("%s: does not fit (%zu B)\n", name, lds); return; }
  const int th = (OH + 3) / 4, tl = (OL + 3) / 4;
  a.ng = std::max((th + a.nth - 1) / a.nth, a.ntl ? (tl + a.ntl - 1) / a.ntl : 0);
  const size_t HWl = (size_t)Hl * Hl;
  a.xh = dalloc((size_t)B * CH * 4 * HWl, 1.f); a.xl = dalloc((size_t)B * a.CL * HWl, 1.f);
  a.yh = dalloc((size_t)B * OH * 4 * HWl, 0.f); a.yl = OL ? dalloc((size_t)B * OL * HWl, 0.f) : nullptr;
  a.wimg = dalloc((size_t)a.ng * a.gimg_floats, 0.1f);
  a.ep_h = dalloc((size_t)(4 * a.ng * a.nth + 4) * 4, 1.f); a.ep_l = dalloc((size_t)(4 * a.ng * std::max(a.ntl, 1) + 4) * 4, 1.f);
  a.dwrec_h = dalloc((size_t)(4 * a.ng * a.nth + 4) * 24, 0.3f); a.dwrec_l = dalloc((size_t)(4 * a.ng * std::max(a.ntl, 1) + 4) * 24, 0.3f);
  const int nblk = 8 * ((B + 7) / 8) * a.ng;
  unsigned long long* st = nullptr;
#ifdef ILB_TIMING
  (void)hipMalloc(&st, (size_t)nblk * 8 * 8);
  (void)hipMemset(st, 0, (size_t)nblk * 8 * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ilb_stamps), &st, sizeof(st));
#endif
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<float> ts;
  for (int it = 0; it < 12; ++it) {
    (void)hipEventRecord(e0);
    const int rc = csn_launch_ilb(a, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    if (rc != 0) { printf("%s: launch failed %d\n", name, rc); return; }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  printf("%-10s B %d [%d,%d]->[%d,%d] @%d^2/%d^2  groups %d  blocks %d x %d threads  LDS %.1f KB: %.1f us per launch (min %.1f)\n", name, B, CH, CL,
         OH, OL, 2 * Hl, Hl, a.ng, nblk, a.nthreads, lds / 1024.0, ts[ts.size() / 2], ts[0]);
#ifdef ILB_TIMING
  std::vector<unsigned long long> h((size_t)nblk * 8);
  (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t5 = 0;
  double ph[5] = {0, 0, 0, 0, 0};
  int n = 0;
  for (int b = 0; b < nblk; ++b) {
    const unsigned long long* s = &h[(size_t)b * 8];
    if (!s[0] || !s[5]) continue;
    t0 = std::min(t0, s[0]); t5 = std::max(t5, s[5]);
    for (int k = 0; k < 5; ++k) ph[k] += (double)(s[k + 1] - s[k]) * 0.01;
    ++n;
  }
  std::vector<double> starts;
  for (int b = 0; b < nblk; ++b) if (h[(size_t)b * 8]) starts.push_back((double)(h[(size_t)b * 8] - t0) * 0.01);
  std::sort(starts.begin(), starts.end());
  printf("           blocks that ran %d; first start -> last end %.1f us; mean per block (us): loads issued + weights + zero %.2f | contraction (all waves) %.2f | z exchange + epilogue %.2f |"
         " dw1 %.2f | dw2 + stores (all waves) %.2f; block start times: median %.1f, p90 %.1f, last %.1f us\n", n, (double)(t5 - t0) * 0.01,
         ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, starts[starts.size() / 2], starts[starts.size() * 9 / 10], starts.back());
#endif
}

int main(int argc, char** argv) {
  if (argc > 1) {   // counter passes (tools/gpu_ilb_pmc.sh): two geometries only
    run("stage3.1", 64, 23, 26, 27, 26, 28, 1);
    run("stage4.1", 64, 18, 31, 31, 27, 14, 1);
    return 0;
  }
  // csnet-L-x2 at 224 x 224, batch 64 (per-block channel plan)
  run("stage3.1", 64, 23, 26, 27, 26, 28, 1);
  run("stage3.3", 64, 25, 20, 17, 21, 28, 1);
  run("stage3.5", 64, 19, 25, 38, 0, 28, 1);
  run("stage4.1", 64, 18, 31, 31, 27, 14, 1);
  run("stage4.2", 64, 31, 27, 26, 44, 14, 1);
  run("stage4.3", 64, 26, 44, 64, 0, 14, 1);
  run("stage4.0 3x3 (1,1)", 64, 38, 0, 18, 31, 14, 1, 1, 1);
  run("stage4.0 3x3 (1,2)", 64, 38, 0, 18, 31, 14, 1, 1, 2);
  run("stage4.1/2", 64, 18, 31, 31, 27, 14, 2);
  run("stage4.3/2", 64, 26, 44, 64, 0, 14, 2);
  return 0;
}
