// wgbf_probe.hip -- wgrad_bf16_kernel (sod100k_amd/csrc/k_wgrad_bf.hip) on its own: lane maps of v_mfma_f32_32x32x16_bf16 for every
// sub-block factor S, multi-source rows, the pooled operand, tails -- against a double-precision host sum -- and the launch time
// of the three big stage-1 passes at batch 256.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../sod100k_amd/csrc -o wgbf_probe wgbf_probe.hip
#include "../../sod100k_amd/csrc/k_wgrad_bf.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static unsigned short f2bf(float f) {
  unsigned u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned rng = 12345u;
static float rnd() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xffff) / 65536.0f - 0.5f; }

struct Case { const char* name; int nrs; int rn[3]; int rctot[3]; int ncs; int cn[3]; int cctot[3]; int H, W, B; bool pool; bool check; bool taps = false; int dils[3] = {1, 1, 1}; };

static int run(const Case& c) {
  const int HW = c.H * c.W;
  const int64_t chw = c.pool ? 4 * (int64_t)HW : HW;
  std::vector<unsigned short> hr[3], hc[3];
  unsigned short *dr[3] = {0, 0, 0}, *dc[3] = {0, 0, 0};
  WgArgs a;
  memset(&a, 0, sizeof(a));
  int R = 0, K = 0;
  for (int q = 0; q < c.nrs; ++q) {
    hr[q].resize((size_t)c.B * c.rctot[q] * HW);
    for (auto& v : hr[q]) v = f2bf(rnd());
    hipMalloc(&dr[q], hr[q].size() * 2);
    hipMemcpy(dr[q], hr[q].data(), hr[q].size() * 2, hipMemcpyHostToDevice);
    // the slice starts at plane 1 when the tensor has spare planes (pointer offsets inside a tensor)
    const int c0 = c.rctot[q] > c.rn[q] ? 1 : 0;
    a.rs[q].ptr = reinterpret_cast<const float*>(dr[q] + (int64_t)c0 * HW); a.rs[q].ctot = c.rctot[q]; a.rs[q].n = c.rn[q];
    R += c.rn[q];
  }
  a.nrs = c.nrs;
  for (int q = 0; q < c.ncs; ++q) {
    hc[q].resize((size_t)c.B * c.cctot[q] * chw);
    for (auto& v : hc[q]) v = f2bf(rnd());
    hipMalloc(&dc[q], hc[q].size() * 2);
    hipMemcpy(dc[q], hc[q].data(), hc[q].size() * 2, hipMemcpyHostToDevice);
    const int c0 = c.cctot[q] > c.cn[q] ? 1 : 0;
    a.ps.src[q].ptr = reinterpret_cast<const float*>(dc[q] + (int64_t)c0 * chw); a.ps.src[q].C = c.cn[q]; a.ps.src[q].Ctot = c.cctot[q];
    a.ps.src[q].mode = c.taps ? PW_TAPS : (c.pool ? PW_POOL2 : PW_OWN); a.ps.src[q].K = (c.taps ? 9 : 1) * c.cn[q];
    a.ps.src[q].dil = c.dils[q];
    K += (c.taps ? 9 : 1) * c.cn[q];
  }
  a.ps.nsrc = c.ncs; a.ps.cin = K; a.ps.nrows = R;
  a.Hr = c.H; a.Wr = c.W; a.B = c.B; a.a16 = 1;
  a.rows16 = (R + 15) & ~15; a.k16 = (K + 15) & ~15;
  if (!(c.taps ? csn_wgrad_bf3_eligible(a) : csn_wgrad_bf_eligible(a))) { printf("%-34s NOT ELIGIBLE\n", c.name); return 1; }
  a.nblk = c.taps ? csn_wgrad_bf3_blocks(a) : csn_wgrad_bf_blocks(a);
  auto launch = [&]() { return c.taps ? csn_launch_wgrad_bf3(a, nullptr) : csn_launch_wgrad_bf(a, nullptr); };
  const size_t pf = (size_t)a.nblk * a.rows16 * a.k16;
  hipMalloc(&a.partial, pf * 4);
  hipMemset(a.partial, 0xff, pf * 4);
  int st = launch();
  hipError_t e = hipDeviceSynchronize();
  if (st != 0 || e != hipSuccess) { printf("%-34s launch failed %d %s\n", c.name, st, hipGetErrorString(e)); return 1; }
  WgBfCfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  if (c.taps) { WgBf3Cfg c3; wgbf3_config(a, &c3); cfg.slog = c3.slog; cfg.ntr = c3.ntr; cfg.ntc = c3.ntc; cfg.L = c3.L; cfg.nitems = c3.nitems; cfg.nblk = c3.nblk; }
  else wgbf_config(a, &cfg);
  float ms = 0.f;
  {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = c.check ? 1 : 5;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
  }
  const double bytes = 2.0 * c.B * HW * R + 2.0 * c.B * chw * (c.taps ? K / 9 : K);
  int bad = 0;
  double maxerr = 0;
  if (c.check) {
    std::vector<float> part(pf);
    hipMemcpy(part.data(), a.partial, pf * 4, hipMemcpyDeviceToHost);
    std::vector<double> ref((size_t)R * K, 0.0), mag((size_t)R * K, 0.0);
    std::vector<float> dzv(R), gv(K);
    if (c.taps) {
      const int Cc = K / 9;
      std::vector<const unsigned short*> cpl(Cc);
      for (int b = 0; b < c.B; ++b) {
        int k = 0;
        std::vector<int> cdil(Cc);
        for (int q = 0; q < c.ncs; ++q) {
          const int c0 = c.cctot[q] > c.cn[q] ? 1 : 0;
          for (int i = 0; i < c.cn[q]; ++i) { cdil[k] = c.dils[q]; cpl[k++] = &hc[q][((size_t)b * c.cctot[q] + c0 + i) * HW]; }
        }
        for (int p = 0; p < HW; ++p) {
          const int y = p / c.W, x = p % c.W;
          int r = 0;
          for (int q = 0; q < c.nrs; ++q) {
            const int c0 = c.rctot[q] > c.rn[q] ? 1 : 0;
            for (int i = 0; i < c.rn[q]; ++i) dzv[r++] = bf2f(hr[q][((size_t)b * c.rctot[q] + c0 + i) * HW + p]);
          }
          for (int ch = 0; ch < Cc; ++ch)
            for (int t = 0; t < 9; ++t) {
              const int yy = y + (t / 3 - 1) * cdil[ch], xx = x + (t % 3 - 1) * cdil[ch];
              if (yy < 0 || yy >= c.H || xx < 0 || xx >= c.W) continue;
              const double v = bf2f(cpl[ch][yy * c.W + xx]);
              for (int i = 0; i < R; ++i) { ref[(size_t)i * K + 9 * ch + t] += dzv[i] * v; mag[(size_t)i * K + 9 * ch + t] += fabs(dzv[i] * v); }
            }
        }
      }
    } else
    for (int b = 0; b < c.B; ++b)
      for (int p = 0; p < HW; ++p) {
        int r = 0;
        for (int q = 0; q < c.nrs; ++q) {
          const int c0 = c.rctot[q] > c.rn[q] ? 1 : 0;
          for (int i = 0; i < c.rn[q]; ++i) dzv[r++] = bf2f(hr[q][((size_t)b * c.rctot[q] + c0 + i) * HW + p]);
        }
        int k = 0;
        for (int q = 0; q < c.ncs; ++q) {
          const int c0 = c.cctot[q] > c.cn[q] ? 1 : 0;
          for (int i = 0; i < c.cn[q]; ++i) {
            const unsigned short* pl = &hc[q][((size_t)b * c.cctot[q] + c0 + i) * chw];
            if (c.pool) {
              const int y = p / c.W, x = p % c.W;
              const unsigned short* t = pl + (size_t)(2 * y) * (2 * c.W) + 2 * x;
              gv[k++] = fmaxf(fmaxf(bf2f(t[0]), bf2f(t[1])), fmaxf(bf2f(t[2 * c.W]), bf2f(t[2 * c.W + 1])));
            } else {
              gv[k++] = bf2f(pl[p]);
            }
          }
        }
        for (int i = 0; i < R; ++i)
          for (int j = 0; j < K; ++j) { ref[(size_t)i * K + j] += (double)dzv[i] * gv[j]; mag[(size_t)i * K + j] += fabs((double)dzv[i] * gv[j]); }
      }
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < K; ++j) {
        double s = 0;
        for (int blk = 0; blk < a.nblk; ++blk) s += part[((size_t)blk * a.rows16 + i) * a.k16 + j];
        const double err = fabs(s - ref[(size_t)i * K + j]) / (mag[(size_t)i * K + j] + 1e-30);
        if (!(err <= 2e-6)) ++bad;
        if (err > maxerr || err != err) maxerr = err;
      }
  }
  printf("%-34s S=%d tiles %dx%d L=%d items %d blocks %d  %8.1f us  %7.1f GB/s  %s max rel err %.2e bad %d\n", c.name, 1 << cfg.slog,
         cfg.ntr, cfg.ntc, cfg.L, cfg.nitems, cfg.nblk, ms * 1e3, bytes / (ms * 1e-3) / 1e9, c.check ? "checked" : "timed  ", maxerr, bad);
  for (int q = 0; q < 3; ++q) { if (dr[q]) hipFree(dr[q]); if (dc[q]) hipFree(dc[q]); }
  hipFree(a.partial);
  return bad;
}

int main() {
  const Case cases[] = {
      {"own 18x13 112^2 S4", 1, {18, 0, 0}, {18, 0, 0}, 1, {13, 0, 0}, {13, 0, 0}, 112, 112, 3, false, true},
      {"own 3src 10+9+10 x 12 56^2", 3, {10, 9, 10}, {10, 11, 10}, 1, {12, 0, 0}, {14, 0, 0}, 56, 56, 3, false, true},
      {"own 27x23 56^2 S2", 1, {27, 0, 0}, {27, 0, 0}, 1, {23, 0, 0}, {23, 0, 0}, 56, 56, 3, false, true},
      {"own 48x64 28^2 S1", 1, {48, 0, 0}, {51, 0, 0}, 1, {64, 0, 0}, {64, 0, 0}, 28, 28, 5, false, true},
      {"own 31x8 (4,1) 48x40", 1, {31, 0, 0}, {31, 0, 0}, 1, {8, 0, 0}, {8, 0, 0}, 48, 40, 2, false, true},
      {"own 7x31 (1,4) 48x40", 1, {7, 0, 0}, {7, 0, 0}, 1, {31, 0, 0}, {31, 0, 0}, 48, 40, 2, false, true},
      {"own 79x1 cls-like 32^2", 1, {48, 0, 0}, {79, 0, 0}, 1, {1, 0, 0}, {1, 0, 0}, 32, 32, 2, false, true},
      {"own 4x4 map", 1, {20, 0, 0}, {20, 0, 0}, 1, {20, 0, 0}, {20, 0, 0}, 4, 4, 2, false, true},
      {"pool 11x13 56^2 S4", 1, {11, 0, 0}, {11, 0, 0}, 1, {13, 0, 0}, {13, 0, 0}, 56, 56, 3, true, true},
      {"pool 22x34 56^2 S2", 1, {22, 0, 0}, {22, 0, 0}, 1, {34, 0, 0}, {36, 0, 0}, 56, 56, 2, true, true},
      {"pool 38x(30+21) 24x16 S1", 1, {38, 0, 0}, {38, 0, 0}, 2, {30, 21, 0}, {30, 21, 0}, 24, 16, 2, true, true},
      {"taps 13x3 112^2 S4", 1, {13, 0, 0}, {13, 0, 0}, 1, {3, 0, 0}, {3, 0, 0}, 112, 112, 2, false, true, true},
      {"taps 28x18 56^2 S1", 1, {28, 0, 0}, {30, 0, 0}, 1, {18, 0, 0}, {18, 0, 0}, 56, 56, 2, false, true, true},
      {"taps 21x(12+18) 56^2 2src", 1, {21, 0, 0}, {21, 0, 0}, 2, {12, 18, 0}, {12, 19, 0}, 56, 56, 2, false, true, true},
      {"taps 23x51 56^2 chunks", 1, {23, 0, 0}, {23, 0, 0}, 1, {51, 0, 0}, {51, 0, 0}, 56, 56, 2, false, true, true},
      {"taps 10x3 48x40 S4", 1, {10, 0, 0}, {10, 0, 0}, 1, {3, 0, 0}, {3, 0, 0}, 48, 40, 3, false, true, true},
      {"taps 40x5 16x8 (3,1)", 1, {40, 0, 0}, {40, 0, 0}, 1, {5, 0, 0}, {5, 0, 0}, 16, 8, 2, false, true, true},
      {"ms 17x(2,1,2) dil 1,2,4 56x48", 1, {17, 0, 0}, {17, 0, 0}, 3, {2, 1, 2}, {8, 8, 8}, 56, 48, 2, false, true, true, {1, 2, 4}},
      {"ms 17x(2,1) dil 8,16 56x48", 1, {17, 0, 0}, {17, 0, 0}, 2, {2, 1, 0}, {8, 8, 0}, 56, 48, 2, false, true, true, {8, 16, 1}},
      {"ms 38x(6,6,7) dil 2,4,8 56^2", 1, {38, 0, 0}, {38, 0, 0}, 3, {6, 6, 7}, {24, 24, 24}, 56, 56, 2, false, true, true, {2, 4, 8}},
      {"ms 38x5 dil 16 56^2", 1, {38, 0, 0}, {38, 0, 0}, 1, {5, 0, 0}, {24, 0, 0}, 56, 56, 2, false, true, true, {16, 1, 1}},
      {"ms 38x(6,6,7) 56^2 B256", 1, {38, 0, 0}, {38, 0, 0}, 3, {6, 6, 7}, {24, 24, 24}, 56, 56, 256, false, false, true, {2, 4, 8}},
      {"ms 17x(2,1,2) 112^2 B256", 1, {17, 0, 0}, {17, 0, 0}, 3, {2, 1, 2}, {8, 8, 8}, 112, 112, 256, false, false, true, {1, 2, 4}},
      {"taps 13x3 224^2 B256", 1, {13, 0, 0}, {13, 0, 0}, 1, {3, 0, 0}, {3, 0, 0}, 224, 224, 256, false, false, true},
      {"taps 28x18 112^2 B256", 1, {28, 0, 0}, {28, 0, 0}, 1, {18, 0, 0}, {18, 0, 0}, 112, 112, 256, false, false, true},
      {"taps 23x51 56^2 B256", 1, {23, 0, 0}, {23, 0, 0}, 1, {51, 0, 0}, {51, 0, 0}, 56, 56, 256, false, false, true},
      {"own 18x13 224^2 B256", 1, {18, 0, 0}, {18, 0, 0}, 1, {13, 0, 0}, {13, 0, 0}, 224, 224, 256, false, false},
      {"own 29x12 112^2 B256", 2, {18, 11, 0}, {18, 11, 0}, 1, {12, 0, 0}, {12, 0, 0}, 112, 112, 256, false, false},
      {"pool 11x13 112^2 B256", 1, {11, 0, 0}, {11, 0, 0}, 1, {13, 0, 0}, {13, 0, 0}, 112, 112, 256, true, false},
      {"own 27x23 56^2 B256", 1, {27, 0, 0}, {27, 0, 0}, 1, {23, 0, 0}, {23, 0, 0}, 56, 56, 256, false, false},
      {"own 48x64 28^2 B256", 1, {48, 0, 0}, {48, 0, 0}, 1, {64, 0, 0}, {64, 0, 0}, 28, 28, 256, false, false},
  };
  int bad = 0;
  for (const Case& c : cases) bad += run(c);
  printf(bad ? "wgbf_probe: FAILED (%d)\n" : "wgbf_probe: ok\n", bad);
  return bad ? 1 : 0;
}
