// csf_head.inl -- host side of the CSF+Res2Net head (include/csf_hip.h); included by csn_plan.hip (shares its
// error plumbing).  The launch sequence is code, not data: csf_walk() visits every step of
// CSFNet.forward after `self.base(x)` (CSF+Res2Net/networks/csf_res2net.py:250-255) in order and either only lays
// out workspace / weight images / preparation jobs (creation) or also launches (forward).
#include "../../include/csf_hip.h"
#include "csf_kernels.h"

struct CsfCopyJob { int64_t src, dst; int n; };          // small parameter vectors: arena -> packed
struct CsfPrepJob { CsfPrepArgs a; int64_t src_off, dst_off; };

struct csf_head {
  csf_head_desc d;
  int B = 0, h[CSF_MAX_BRANCH] = {0}, w[CSF_MAX_BRANCH] = {0}, oh = 0, ow = 0;
  float* packed = nullptr;               // weight images + parameter vectors (library owned)
  int64_t packed_floats = 0;
  size_t ws_bytes = 0;
  int64_t macs = 0;
  bool refreshed = false;
  std::vector<CsfPrepJob> prep;
  std::vector<CsfCopyJob> copies;
  struct Stage { int64_t off; int C, H, W; } stage[3][CSF_MAX_BRANCH];
};

namespace {

struct CsfWalk {
  csf_head* H;
  const float* const* feats = nullptr;
  float* logits = nullptr;
  char* ws = nullptr;
  void* stream = nullptr;
  bool dry = true;                       // creation: layout + jobs only
  int64_t ws_top = 0, pk_top = 0;
  int64_t macs = 0;

  int64_t ws_alloc(int64_t bytes) { const int64_t o = ws_top; ws_top += align_up(bytes, 256); return o; }
  int64_t pk_alloc(int64_t floats) { const int64_t o = pk_top; pk_top += align_up(floats, 64); return o; }
  float* wsf(int64_t off) const { return reinterpret_cast<float*>(ws + off); }
  float* pk(int64_t off) const { return H->packed + off; }
  int64_t vec(int64_t arena_off, int n) {   // parameter vector copied into the packed buffer at refresh
    const int64_t o = pk_alloc(n);
    if (dry) H->copies.push_back(CsfCopyJob{arena_off, o, n});
    return o;
  }
};

struct CsfSrc {          // one K segment of a GEMM
  const float* ptr;      // null during the dry walk
  int64_t tensor_floats; // extent from ptr to the end of the tensor
  int C, ctot, Hs, Ws, mode;
};

int csf_gemm(CsfWalk& W, int M, int nseg, const CsfSrc* src, int taps, int dil, int64_t w_arena, int ld,
             const int* col0, float* out, int out_ctot, int Ho, int Wo) {
  csf_head* H = W.H;
  const int mt = M >= 48 ? 4 : 2, BM = 16 * mt;
  const int Mp = (M + BM - 1) / BM * BM;
  int Kp = 0;
  CsfPrepJob pj{};
  CsfGemmArgs a{};
  a.taps = taps;
  a.dil = dil;
  a.nseg = nseg;
  int64_t kreal = 0;
  for (int s = 0; s < nseg; ++s) {
    const int chunks = (src[s].C + CSF_KC - 1) / CSF_KC;
    if (src[s].tensor_floats * 4 >= ((int64_t)1 << 31)) FAIL(CSN_E_UNSUPPORTED, "csf: tensor of 2 GiB or more");
    CsfSeg& g = a.seg[s];
    g.src = src[s].ptr;
    g.bytes = (unsigned)(src[s].tensor_floats * 4);
    g.Hs = src[s].Hs;
    g.Ws = src[s].Ws;
    g.cstride = g.Hs * g.Ws;
    g.nstride = src[s].ctot * g.cstride;
    g.chunks = chunks;
    g.mode = src[s].mode;
    g.ry = (float)g.Hs / (float)Ho;
    g.rx = (float)g.Ws / (float)Wo;
    if (taps) {
      pj.a.seg[0] = CsfPrepSeg{chunks * CSF_KC, src[s].C, 0};
      Kp = 9 * chunks * CSF_KC;
      kreal = 9 * (int64_t)src[s].C;
    } else {
      pj.a.seg[s] = CsfPrepSeg{Kp, src[s].C, col0[s]};
      Kp += chunks * CSF_KC;
      kreal += src[s].C;
    }
  }
  const int64_t img = W.pk_alloc((int64_t)Mp * Kp);
  const int Ntot = H->B * Ho * Wo;
  W.macs += (int64_t)M * kreal * Ntot;
  if (W.dry) {
    pj.a.M = M; pj.a.Mp = Mp; pj.a.Kp = Kp; pj.a.ld = ld; pj.a.taps = taps; pj.a.nseg = nseg;
    pj.src_off = w_arena;
    pj.dst_off = img;
    H->prep.push_back(pj);
    return CSN_OK;
  }
  a.A = W.pk(img);
  a.M = M;
  a.Kp = Kp;
  a.out = out;
  a.out_nstride = (long long)out_ctot * Ho * Wo;
  a.Ho = Ho; a.Wo = Wo; a.HWo = Ho * Wo; a.Ntot = Ntot;
  a.n_mtiles = Mp / BM;
  a.n_ntiles = (Ntot + CSF_BN - 1) / CSF_BN;
  LAUNCH_TRY(csf_launch_gemm(a, mt, W.stream));
  return CSN_OK;
}

// combine (+ optional up-sampled partials) -> GroupNorm statistics -> scale/shift tables; returns their packed offsets
int csf_group_norm(CsfWalk& W, float* s, int C, int Hh, int Ww, int nz, const CsfZ* z, const csf_gn_off& gn,
                   int64_t* scale_off, int64_t* shift_off, int64_t* alpha_off) {
  csf_head* H = W.H;
  const int groups = H->d.gn_groups, cpg = C / groups, HW = Hh * Ww;
  const int glen = cpg * HW;
  const int nslab = std::max(1, std::min(64, (glen + 8191) / 8192));
  const int slab_len = (glen + nslab - 1) / nslab;
  const int64_t part = W.ws_alloc((int64_t)H->B * groups * nslab * 2 * sizeof(double));
  const int64_t sc = W.ws_alloc((int64_t)H->B * C * 4), sh = W.ws_alloc((int64_t)H->B * C * 4);
  const int64_t gamma = W.vec(gn.weight, C), beta = W.vec(gn.bias, C), alpha = W.vec(gn.prelu, C);
  *scale_off = sc; *shift_off = sh; *alpha_off = alpha;
  if (W.dry) return CSN_OK;
  CsfCombArgs c{};
  c.s = s; c.B = H->B; c.C = C; c.H = Hh; c.W = Ww; c.HW = HW; c.cpg = cpg; c.groups = groups;
  c.nz = nz;
  for (int i = 0; i < nz; ++i) c.z[i] = z[i];
  c.part = reinterpret_cast<double*>(W.ws + part);
  c.nslab = nslab; c.slab_len = slab_len;
  LAUNCH_TRY(csf_launch_combine(c, W.stream));
  CsfGnFinArgs f{};
  f.part = c.part; f.nslab = nslab; f.cpg = cpg; f.groups = groups; f.C = C; f.HW = HW; f.B = H->B;
  f.gamma = W.pk(gamma); f.beta = W.pk(beta); f.eps = 1e-5f;
  f.scale = W.wsf(sc); f.shift = W.wsf(sh);
  LAUNCH_TRY(csf_launch_gn_finalize(f, W.stream));
  return CSN_OK;
}

int csf_apply(CsfWalk& W, float* s, int C, int HW, int64_t sc, int64_t sh, int64_t alpha) {
  if (W.dry) return CSN_OK;
  CsfApplyArgs a{};
  a.s = s; a.scale = W.wsf(sc); a.shift = W.wsf(sh); a.alpha = W.pk(alpha);
  a.C = C; a.HW = HW; a.total = (long long)W.H->B * C * HW;
  LAUNCH_TRY(csf_launch_apply(a, W.stream));
  return CSN_OK;
}

#define CSF_TRY(expr) do { int _s = (expr); if (_s != CSN_OK) return _s; } while (0)

int csf_walk(CsfWalk& W) {
  csf_head* H = W.H;
  const csf_head_desc& d = H->d;
  const int nb = d.n_branch, B = H->B;
  int bi[CSF_MAX_BRANCH + 1] = {0}, bo[CSF_MAX_BRANCH + 1] = {0}, HW[CSF_MAX_BRANCH];
  for (int i = 0; i < nb; ++i) {
    bi[i + 1] = bi[i] + d.cin[i];
    bo[i + 1] = bo[i] + d.cmid[i];
    HW[i] = H->h[i] * H->w[i];
  }
  const int Tin = bi[nb], Tm = bo[nb];
  auto feat = [&](int i) -> const float* { return W.dry ? nullptr : W.feats[i]; };
  auto wsp = [&](int64_t off) -> float* { return W.dry ? nullptr : W.wsf(off); };

  // ---- fuse: gOctaveCBR 4 -> 4, 1x1 (gOctConv.py:60-114).  Output branch i at its own resolution takes the branches
  // i' <= i (resized down first, 99-101) in ONE contraction; its contributions to the finer branches j < i are
  // contracted at resolution i (rows 0..bo[i]) and up-sampled afterwards (96-98) by the combine pass.
  int64_t S[CSF_MAX_BRANCH], Z[CSF_MAX_BRANCH] = {0};
  for (int i = 0; i < nb; ++i) S[i] = W.ws_alloc((int64_t)B * d.cmid[i] * HW[i] * 4);
  for (int i = 1; i < nb; ++i) Z[i] = W.ws_alloc((int64_t)B * bo[i] * HW[i] * 4);
  for (int i = 0; i < nb; ++i) {
    CsfSrc src[CSF_MAX_SEG];
    int col0[CSF_MAX_SEG];
    for (int k = 0; k <= i; ++k) {
      src[k] = CsfSrc{feat(k), (int64_t)B * d.cin[k] * HW[k], d.cin[k], d.cin[k], H->h[k], H->w[k], k == i ? CSF_OWN : CSF_RESIZE};
      col0[k] = bi[k];
    }
    CSF_TRY(csf_gemm(W, d.cmid[i], i + 1, src, 0, 1, d.fuse_w + (int64_t)bo[i] * Tin, Tin, col0, wsp(S[i]), d.cmid[i],
                     H->h[i], H->w[i]));
    if (i >= 1) {
      CsfSrc own{feat(i), (int64_t)B * d.cin[i] * HW[i], d.cin[i], d.cin[i], H->h[i], H->w[i], CSF_OWN};
      const int c0 = bi[i];
      CSF_TRY(csf_gemm(W, bo[i], 1, &own, 0, 1, d.fuse_w, Tin, &c0, wsp(Z[i]), bo[i], H->h[i], H->w[i]));
    }
  }
  for (int j = 0; j < nb; ++j) {
    CsfZ z[3];
    int nz = 0;
    for (int i = j + 1; i < nb; ++i, ++nz) {
      z[nz].z = W.dry ? nullptr : W.wsf(Z[i]) + (int64_t)bo[j] * HW[i];
      z[nz].nstride = (long long)bo[i] * HW[i];
      z[nz].Hz = H->h[i]; z[nz].Wz = H->w[i];
      z[nz].ry = (float)H->h[i] / (float)H->h[j];
      z[nz].rx = (float)H->w[i] / (float)H->w[j];
    }
    int64_t sc, sh, al;
    CSF_TRY(csf_group_norm(W, wsp(S[j]), d.cmid[j], H->h[j], H->w[j], nz, z, d.fuse_gn[j], &sc, &sh, &al));
    CSF_TRY(csf_apply(W, wsp(S[j]), d.cmid[j], HW[j], sc, sh, al));
    H->stage[0][j] = {S[j], d.cmid[j], H->h[j], H->w[j]};
  }

  // ---- ms: PallMSBlock (csf_res2net.py:174-223): five dense dilated 3x3 convolutions per branch write channel
  // slices of one tensor (torch.cat, 212), then GroupNorm + PReLU
  static const int dil[CSF_NDIL] = {1, 2, 4, 8, 16};
  int64_t Mo[CSF_MAX_BRANCH];
  for (int j = 0; j < nb; ++j) Mo[j] = W.ws_alloc((int64_t)B * d.cmid[j] * HW[j] * 4);
  for (int j = 0; j < nb; ++j) {
    int row = 0;
    for (int k = 0; k < CSF_NDIL; ++k) {
      const int co = d.ms_split[j][k];
      if (co <= 0) continue;
      CsfSrc src{wsp(S[j]), (int64_t)B * d.cmid[j] * HW[j], d.cmid[j], d.cmid[j], H->h[j], H->w[j], CSF_OWN};
      float* out = W.dry ? nullptr : W.wsf(Mo[j]) + (int64_t)row * HW[j];
      CSF_TRY(csf_gemm(W, co, 1, &src, 9, dil[k], d.ms_w[j][k], d.cmid[j] * 9, nullptr, out, d.cmid[j], H->h[j], H->w[j]));
      row += co;
    }
    int64_t sc, sh, al;
    CSF_TRY(csf_group_norm(W, wsp(Mo[j]), d.cmid[j], H->h[j], H->w[j], 0, nullptr, d.ms_gn[j], &sc, &sh, &al));
    CSF_TRY(csf_apply(W, wsp(Mo[j]), d.cmid[j], HW[j], sc, sh, al));
    H->stage[1][j] = {Mo[j], d.cmid[j], H->h[j], H->w[j]};
  }

  // ---- fuse1x1: gOctaveCBR 4 -> 1 (csf_res2net.py:243-244): branch 0 directly, branches i >= 1 contracted at their
  // own resolution and up-sampled by the combine pass
  const int64_t F = W.ws_alloc((int64_t)B * Tm * HW[0] * 4);
  int64_t Z1[CSF_MAX_BRANCH] = {0};
  for (int i = 1; i < nb; ++i) Z1[i] = W.ws_alloc((int64_t)B * Tm * HW[i] * 4);
  for (int i = 0; i < nb; ++i) {
    CsfSrc src{wsp(Mo[i]), (int64_t)B * d.cmid[i] * HW[i], d.cmid[i], d.cmid[i], H->h[i], H->w[i], CSF_OWN};
    const int c0 = bo[i];
    CSF_TRY(csf_gemm(W, Tm, 1, &src, 0, 1, d.fuse1_w, Tm, &c0, wsp(i == 0 ? F : Z1[i]), Tm, H->h[i], H->w[i]));
  }
  {
    CsfZ z[3];
    int nz = 0;
    for (int i = 1; i < nb; ++i, ++nz) {
      z[nz].z = wsp(Z1[i]);
      z[nz].nstride = (long long)Tm * HW[i];
      z[nz].Hz = H->h[i]; z[nz].Wz = H->w[i];
      z[nz].ry = (float)H->h[i] / (float)H->h[0];
      z[nz].rx = (float)H->w[i] / (float)H->w[0];
    }
    int64_t sc, sh, al;
    CSF_TRY(csf_group_norm(W, wsp(F), Tm, H->h[0], H->w[0], nz, z, d.fuse1_gn, &sc, &sh, &al));
    H->stage[2][0] = {F, Tm, H->h[0], H->w[0]};
    // ---- cls_layer on PReLU(GroupNorm(.)) + resize to the input size (csf_res2net.py:253-254)
    const int64_t lo = W.ws_alloc((int64_t)B * HW[0] * 4);
    const int64_t cw = W.vec(d.cls_w, Tm), cb = W.vec(d.cls_b, 1);
    if (!W.dry) {
      CsfClsArgs c{};
      c.s = W.wsf(F); c.scale = W.wsf(sc); c.shift = W.wsf(sh); c.alpha = W.pk(al);
      c.w = W.pk(cw); c.bias = W.pk(cb); c.out = W.wsf(lo);
      c.C = Tm; c.HW = HW[0]; c.B = B;
      LAUNCH_TRY(csf_launch_cls(c, W.stream));
      CsfResizeArgs r{};
      r.in = W.wsf(lo); r.out = W.logits; r.planes = B; r.Hi = H->h[0]; r.Wi = H->w[0]; r.Ho = H->oh; r.Wo = H->ow;
      r.ry = (float)r.Hi / (float)r.Ho;
      r.rx = (float)r.Wi / (float)r.Wo;
      LAUNCH_TRY(csf_launch_resize(r, W.stream));
    }
  }
  return CSN_OK;
}

}  // namespace

extern "C" {

int csf_head_create(const csf_head_desc* desc, int32_t batch, const int32_t* h, const int32_t* w, int32_t out_h,
                    int32_t out_w, csf_head** out) {
  g_why.clear();
  g_hip_err.clear();
  if (!desc || !h || !w || !out) FAIL(CSN_E_INVALID, "null argument");
  const csf_head_desc& d = *desc;
  if (d.n_branch < 1 || d.n_branch > CSF_MAX_BRANCH || batch < 1 || out_h < 1 || out_w < 1) FAIL(CSN_E_INVALID, "geometry");
  if (d.gn_groups < 1) FAIL(CSN_E_INVALID, "gn_groups");
  int tm = 0;
  for (int i = 0; i < d.n_branch; ++i) {
    if (h[i] < 1 || w[i] < 1 || d.cin[i] < 1 || d.cmid[i] < 1) FAIL(CSN_E_INVALID, "branch geometry");
    if (d.cmid[i] % d.gn_groups) FAIL(CSN_E_INVALID, "GroupNorm: channels not divisible by the group count");
    int sum = 0;
    for (int k = 0; k < CSF_NDIL; ++k) {
      if (d.ms_split[i][k] < 0 || (d.ms_split[i][k] > 0 && d.ms_w[i][k] < 0)) FAIL(CSN_E_INVALID, "ms_split / ms_w");
      sum += d.ms_split[i][k];
    }
    if (sum != d.cmid[i]) FAIL(CSN_E_INVALID, "ms_split does not add up to cmid");
    tm += d.cmid[i];
  }
  if (tm % d.gn_groups) FAIL(CSN_E_INVALID, "GroupNorm(fuse1x1): channels not divisible by the group count");
  csf_head* H = new (std::nothrow) csf_head();
  if (!H) return CSN_E_NOMEM;
  H->d = d;
  H->B = batch;
  for (int i = 0; i < d.n_branch; ++i) { H->h[i] = h[i]; H->w[i] = w[i]; }
  H->oh = out_h;
  H->ow = out_w;
  CsfWalk W;
  W.H = H;
  W.dry = true;
  const int st = csf_walk(W);
  if (st != CSN_OK) { delete H; return st; }
  H->ws_bytes = (size_t)W.ws_top;
  H->packed_floats = W.pk_top;
  H->macs = W.macs;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&H->packed), (size_t)H->packed_floats * sizeof(float));
  if (e != hipSuccess) { delete H; return hip_fail(e, "hipMalloc(weight images)"); }
  *out = H;
  return CSN_OK;
}

void csf_head_destroy(csf_head* H) {
  if (!H) return;
  if (H->packed) (void)hipFree(H->packed);
  delete H;
}

size_t csf_head_workspace_bytes(const csf_head* H) { return H ? H->ws_bytes : 0; }
int64_t csf_head_macs(const csf_head* H) { return H ? H->macs : 0; }

int csf_head_refresh_params(csf_head* H, const float* arena, int64_t arena_floats, void* stream) {
  if (!H || !arena) FAIL(CSN_E_INVALID, "null argument");
  for (const CsfPrepJob& j : H->prep) {
    if (j.src_off < 0 || j.src_off + (int64_t)j.a.M * j.a.ld > arena_floats) FAIL(CSN_E_INVALID, "weight offset outside the arena");
    CsfPrepArgs a = j.a;
    a.src = arena + j.src_off;
    a.dst = H->packed + j.dst_off;
    LAUNCH_TRY(csf_launch_prep(a, stream));
  }
  for (const CsfCopyJob& c : H->copies) {
    if (c.src < 0 || c.src + c.n > arena_floats) FAIL(CSN_E_INVALID, "parameter offset outside the arena");
    HIP_TRY(hipMemcpyAsync(H->packed + c.dst, arena + c.src, (size_t)c.n * sizeof(float), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
  }
  H->refreshed = true;
  return CSN_OK;
}

int csf_head_forward(csf_head* H, const float* const* features, float* logits, void* workspace, void* stream) {
  if (!H || !features || !logits || !workspace) FAIL(CSN_E_INVALID, "null argument");
  if (!H->refreshed) FAIL(CSN_E_STATE, "csf_head_forward before csf_head_refresh_params");
  for (int i = 0; i < H->d.n_branch; ++i)
    if (!features[i]) FAIL(CSN_E_INVALID, "null feature pointer");
  CsfWalk W;
  W.H = H;
  W.feats = features;
  W.logits = logits;
  W.ws = static_cast<char*>(workspace);
  W.stream = stream;
  W.dry = false;
  return csf_walk(W);
}

int csf_head_stage_info(const csf_head* H, int32_t stage, int32_t branch, int64_t* ws_offset_bytes, int32_t* channels,
                        int32_t* height, int32_t* width) {
  if (!H || stage < 0 || stage > 2 || branch < 0 || branch >= (stage == 2 ? 1 : H->d.n_branch)) FAIL(CSN_E_INVALID, "stage / branch");
  const csf_head::Stage& s = H->stage[stage][branch];
  if (ws_offset_bytes) *ws_offset_bytes = s.off;
  if (channels) *channels = s.C;
  if (height) *height = s.H;
  if (width) *width = s.W;
  return CSN_OK;
}

}  // extern "C"
