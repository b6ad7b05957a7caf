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
  int z_lds_max = 12 * 1024;             // floats; CSF_Z_LDS_MAX overrides (tests force the global-memory tap path)
  std::vector<CsfPrepJob> prep;
  std::vector<CsfCopyJob> copies;
  CsfPrepJobDev* jobs_dev = nullptr;     // prep + copies as one device-resident table (one launch per refresh)
  int njobs = 0, job_blocks = 0;
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

struct CsfSubPlan { int M, dil; int64_t w_arena; float* out; };

struct CsfGeom {         // tiling of one launch (known before its output is allocated: split-K needs planes)
  int mt = 4, BM = 64, n_ntiles = 0, tiles = 0, nchunks = 0, ksplit = 1, cps = 0;
  int n_mtiles[CSF_MAX_SUB] = {0};
};

// Under-filled launches (coarse levels: few pixel tiles, K up to 4608) are cut along K until ~3 blocks per CU exist;
// every slice keeps >= 16 chunks.  The planes are summed in slice order by the combine pass (deterministic).
CsfGeom csf_geom(int nsub, const int* M, int nseg, const int* segC, int taps, int Ntot, bool allow_split) {
  CsfGeom g;
  int maxM = 0;
  for (int i = 0; i < nsub; ++i) maxM = std::max(maxM, M[i]);
  // 64-row blocks (3 resident blocks per CU); 32 rows for the 25..28-row dilation groups of the 128-channel MSBlock.
  // 112 / 128-row variants were measured: fewer re-gathers of B, but 2 blocks per CU -- no faster (profiles/r1_notes.md)
  g.mt = maxM >= 48 ? 4 : 2;
  // round 6, csf_gemm3_kernel (fp32 operands as three bfloat16 parts): the split of a gathered B element is vector work that every
  // row of the block shares -- 128-row blocks where M fills them (same lease: head 6.72 -> 6.57 ms)
  if (!csf_gemm_f32() && maxM >= 112) g.mt = 8;
  g.BM = 16 * g.mt;
  g.n_ntiles = (Ntot + CSF_BN - 1) / CSF_BN;
  for (int i = 0; i < nsub; ++i) {
    g.n_mtiles[i] = (M[i] + g.BM - 1) / g.BM;
    g.tiles += g.n_mtiles[i] * g.n_ntiles;
  }
  for (int s = 0; s < nseg; ++s) g.nchunks += (segC[s] + CSF_KC - 1) / CSF_KC;
  if (taps) g.nchunks *= 9;
  if (allow_split)
    while (g.tiles * g.ksplit < 768 && g.ksplit < 8 && g.nchunks / (g.ksplit * 2) >= 16) g.ksplit *= 2;
  g.cps = (g.nchunks + g.ksplit - 1) / g.ksplit;
  return g;
}

int csf_gemm(CsfWalk& W, const CsfGeom& G, int nsub, const CsfSubPlan* subs, int nseg, const CsfSrc* src, int taps,
             int ld, const int* col0, int out_ctot, int Ho, int Wo, int64_t split_stride) {
  csf_head* H = W.H;
  int Kp = 0;
  CsfPrepJob pj{};
  CsfGemmArgs a{};
  a.taps = taps;
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
  const int Ntot = H->B * Ho * Wo;
  int tile0 = 0;
  for (int i = 0; i < nsub; ++i) {
    const int Mp = G.n_mtiles[i] * G.BM;
    const int64_t img = W.pk_alloc((int64_t)Mp * Kp);
    W.macs += (int64_t)subs[i].M * kreal * Ntot;
    if (W.dry) {
      pj.a.M = subs[i].M; pj.a.Mp = Mp; pj.a.Kp = Kp; pj.a.ld = ld; pj.a.taps = taps; pj.a.nseg = nseg;
      pj.src_off = subs[i].w_arena;
      pj.dst_off = img;
      H->prep.push_back(pj);
    }
    a.sub[i] = CsfSub{W.dry ? nullptr : W.pk(img), subs[i].out, subs[i].M, subs[i].dil, G.n_mtiles[i], tile0};
    tile0 += G.n_mtiles[i] * G.n_ntiles * G.ksplit;
  }
  if (W.dry) return CSN_OK;
  a.nsub = nsub;
  a.total_tiles = tile0;
  a.ksplit = G.ksplit;
  a.chunks_per_split = G.cps;
  a.split_stride = split_stride;
  a.Kp = Kp;
  a.out_nstride = (long long)out_ctot * Ho * Wo;
  a.Ho = Ho; a.Wo = Wo; a.HWo = Ho * Wo; a.Ntot = Ntot;
  a.n_ntiles = G.n_ntiles;
  LAUNCH_TRY(csf_launch_gemm(a, G.mt, W.stream));
  return CSN_OK;
}

// combine (split-K planes + optional up-sampled partials) -> GroupNorm statistics -> scale/shift tables
int csf_group_norm(CsfWalk& W, float* s, int ns, int64_t split_stride, int C, int Hh, int Ww, int nz, const CsfZ* z,
                   const csf_gn_off& gn, int64_t* scale_off, int64_t* shift_off, int64_t* alpha_off) {
  csf_head* H = W.H;
  const int groups = H->d.gn_groups, cpg = C / groups, HW = Hh * Ww;
  const int nslab = cpg;                 // one partial per channel plane
  const int64_t part = W.ws_alloc((int64_t)H->B * C * 2 * sizeof(double));
  const int64_t sc = W.ws_alloc((int64_t)H->B * C * 4), sh = W.ws_alloc((int64_t)H->B * C * 4);
  const int64_t gamma = W.vec(gn.weight, C), beta = W.vec(gn.bias, C), alpha = W.vec(gn.prelu, C);
  *scale_off = sc; *shift_off = sh; *alpha_off = alpha;
  if (W.dry) return CSN_OK;
  if (H->B > 65535) FAIL(CSN_E_UNSUPPORTED, "csf: batch exceeds the grid limit");
  CsfCombArgs c{};
  c.s = s; c.ns = ns; c.split_stride = split_stride;
  c.B = H->B; c.C = C; c.H = Hh; c.W = Ww; c.HW = HW; c.cpg = cpg; c.groups = groups;
  c.nz = nz;
  for (int i = 0; i < nz; ++i) c.z[i] = z[i];
  c.part = reinterpret_cast<double*>(W.ws + part);
  for (int i = 0; i < nz; ++i) c.z_floats += z[i].Hz * z[i].Wz;
  c.z_in_lds = c.z_floats > 0 && 4 * c.z_floats <= H->z_lds_max;   // <= 48 KB next to the reduction scratch
  c.step_x = CSN_BLOCK % Ww;
  c.step_y = CSN_BLOCK / Ww;
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
  CsfGeom Gs[CSF_MAX_BRANCH], Gz[CSF_MAX_BRANCH];
  for (int i = 0; i < nb; ++i) {
    const int segc[2] = {i >= 1 ? bi[i] : d.cin[0], d.cin[i]};
    Gs[i] = csf_geom(1, &d.cmid[i], i >= 1 ? 2 : 1, segc, 0, B * HW[i], true);
    S[i] = W.ws_alloc((int64_t)Gs[i].ksplit * B * d.cmid[i] * HW[i] * 4);
  }
  for (int i = 1; i < nb; ++i) {
    Gz[i] = csf_geom(1, &bo[i], 1, &d.cin[i], 0, B * HW[i], true);
    Z[i] = W.ws_alloc((int64_t)Gz[i].ksplit * B * bo[i] * HW[i] * 4);
  }
  // the finer branches i' < i enter branch i's contraction resized to its resolution (F.interpolate BEFORE the conv,
  // gOctConv.py:99-101): done once per level into R[i] = [B][bi[i]][HW[i]] (the GEMM re-gathers its B operand per row
  // tile and K slice -- 8 x 4 times for the coarsest level -- so resizing inside the gather cost 4 loads + a lerp each time)
  int64_t R[CSF_MAX_BRANCH] = {0};
  for (int i = 1; i < nb; ++i) R[i] = W.ws_alloc((int64_t)B * bi[i] * HW[i] * 4);
  for (int i = 1; i < nb; ++i)
    for (int k = 0; k < i; ++k) {
      if (W.dry) continue;
      CsfResizeArgs r{};
      r.in = W.feats[k]; r.out = W.wsf(R[i]) + (int64_t)bi[k] * HW[i];
      r.planes = B * d.cin[k]; r.cpi = d.cin[k]; r.out_nstride = (long long)bi[i] * HW[i];
      r.Hi = H->h[k]; r.Wi = H->w[k]; r.Ho = H->h[i]; r.Wo = H->w[i];
      r.ry = (float)r.Hi / (float)r.Ho;
      r.rx = (float)r.Wi / (float)r.Wo;
      LAUNCH_TRY(csf_launch_resize(r, W.stream));
    }
  for (int i = 0; i < nb; ++i) {
    CsfSrc src[2];
    int col0[2], ns = 0;
    if (i >= 1) {
      src[ns] = CsfSrc{wsp(R[i]), (int64_t)B * bi[i] * HW[i], bi[i], bi[i], H->h[i], H->w[i], CSF_OWN};
      col0[ns++] = 0;
    }
    src[ns] = CsfSrc{feat(i), (int64_t)B * d.cin[i] * HW[i], d.cin[i], d.cin[i], H->h[i], H->w[i], CSF_OWN};
    col0[ns++] = bi[i];
    CsfSubPlan own{d.cmid[i], 1, d.fuse_w + (int64_t)bo[i] * Tin, wsp(S[i])};
    CSF_TRY(csf_gemm(W, Gs[i], 1, &own, ns, src, 0, Tin, col0, d.cmid[i], H->h[i], H->w[i],
                     (int64_t)B * d.cmid[i] * HW[i]));
    if (i >= 1) {
      CsfSrc me{feat(i), (int64_t)B * d.cin[i] * HW[i], d.cin[i], d.cin[i], H->h[i], H->w[i], CSF_OWN};
      CsfSubPlan zp{bo[i], 1, d.fuse_w, wsp(Z[i])};
      const int c0 = bi[i];
      CSF_TRY(csf_gemm(W, Gz[i], 1, &zp, 1, &me, 0, Tin, &c0, bo[i], H->h[i], H->w[i], (int64_t)B * bo[i] * HW[i]));
    }
  }
  for (int j = 0; j < nb; ++j) {
    CsfZ z[3];
    int nz = 0;
    for (int i = j + 1; i < nb; ++i, ++nz) {
      z[nz].z = W.dry ? nullptr : W.wsf(Z[i]) + (int64_t)bo[j] * HW[i];
      z[nz].nstride = (long long)bo[i] * HW[i];
      z[nz].ns = Gz[i].ksplit;
      z[nz].split_stride = (long long)B * bo[i] * HW[i];
      z[nz].Hz = H->h[i]; z[nz].Wz = H->w[i];
      z[nz].ry = (float)H->h[i] / (float)H->h[j];
      z[nz].rx = (float)H->w[i] / (float)H->w[j];
    }
    int64_t sc, sh, al;
    CSF_TRY(csf_group_norm(W, wsp(S[j]), Gs[j].ksplit, (int64_t)B * d.cmid[j] * HW[j], d.cmid[j], H->h[j], H->w[j], nz, z,
                           d.fuse_gn[j], &sc, &sh, &al));
    CSF_TRY(csf_apply(W, wsp(S[j]), d.cmid[j], HW[j], sc, sh, al));
    H->stage[0][j] = {S[j], d.cmid[j], H->h[j], H->w[j]};
  }

  // ---- ms: PallMSBlock (csf_res2net.py:174-223): five dense dilated 3x3 convolutions per branch write channel
  // slices of one tensor (torch.cat, 212) -- ONE launch with five sub-problems -- then GroupNorm + PReLU
  static const int dil[CSF_NDIL] = {1, 2, 4, 8, 16};
  int64_t Mo[CSF_MAX_BRANCH];
  CsfGeom Gm[CSF_MAX_BRANCH];
  for (int j = 0; j < nb; ++j) {
    int Ms[CSF_NDIL], ns = 0;
    for (int k = 0; k < CSF_NDIL; ++k)
      if (d.ms_split[j][k] > 0) Ms[ns++] = d.ms_split[j][k];
    Gm[j] = csf_geom(ns, Ms, 1, &d.cmid[j], 9, B * HW[j], true);
    Mo[j] = W.ws_alloc((int64_t)Gm[j].ksplit * B * d.cmid[j] * HW[j] * 4);
  }
  for (int j = 0; j < nb; ++j) {
    CsfSubPlan subs[CSF_NDIL];
    int ns = 0, row = 0;
    for (int k = 0; k < CSF_NDIL; ++k) {
      const int co = d.ms_split[j][k];
      if (co <= 0) continue;
      subs[ns++] = CsfSubPlan{co, dil[k], d.ms_w[j][k], W.dry ? nullptr : W.wsf(Mo[j]) + (int64_t)row * HW[j]};
      row += co;
    }
    CsfSrc src{wsp(S[j]), (int64_t)B * d.cmid[j] * HW[j], d.cmid[j], d.cmid[j], H->h[j], H->w[j], CSF_OWN};
    CSF_TRY(csf_gemm(W, Gm[j], ns, subs, 1, &src, 9, d.cmid[j] * 9, nullptr, d.cmid[j], H->h[j], H->w[j],
                     (int64_t)B * d.cmid[j] * HW[j]));
    int64_t sc, sh, al;
    CSF_TRY(csf_group_norm(W, wsp(Mo[j]), Gm[j].ksplit, (int64_t)B * d.cmid[j] * HW[j], d.cmid[j], H->h[j], H->w[j], 0,
                           nullptr, d.ms_gn[j], &sc, &sh, &al));
    CSF_TRY(csf_apply(W, wsp(Mo[j]), d.cmid[j], HW[j], sc, sh, al));
    H->stage[1][j] = {Mo[j], d.cmid[j], H->h[j], H->w[j]};
  }

  // ---- fuse1x1: gOctaveCBR 4 -> 1 (csf_res2net.py:243-244): branch 0 directly, branches i >= 1 contracted at their
  // own resolution and up-sampled by the combine pass
  int64_t Z1[CSF_MAX_BRANCH] = {0};
  CsfGeom Gf[CSF_MAX_BRANCH];
  for (int i = 0; i < nb; ++i) {
    Gf[i] = csf_geom(1, &Tm, 1, &d.cmid[i], 0, B * HW[i], i > 0);
    Z1[i] = W.ws_alloc((int64_t)Gf[i].ksplit * B * Tm * HW[i] * 4);
  }
  const int64_t F = Z1[0];
  for (int i = 0; i < nb; ++i) {
    CsfSrc src{wsp(Mo[i]), (int64_t)B * d.cmid[i] * HW[i], d.cmid[i], d.cmid[i], H->h[i], H->w[i], CSF_OWN};
    CsfSubPlan sp{Tm, 1, d.fuse1_w, wsp(Z1[i])};
    const int c0 = bo[i];
    CSF_TRY(csf_gemm(W, Gf[i], 1, &sp, 1, &src, 0, Tm, &c0, Tm, H->h[i], H->w[i], (int64_t)B * Tm * HW[i]));
  }
  {
    CsfZ z[3];
    int nz = 0;
    for (int i = 1; i < nb; ++i, ++nz) {
      z[nz].z = wsp(Z1[i]);
      z[nz].nstride = (long long)Tm * HW[i];
      z[nz].ns = Gf[i].ksplit;
      z[nz].split_stride = (long long)B * Tm * HW[i];
      z[nz].Hz = H->h[i]; z[nz].Wz = H->w[i];
      z[nz].ry = (float)H->h[i] / (float)H->h[0];
      z[nz].rx = (float)H->w[i] / (float)H->w[0];
    }
    int64_t sc, sh, al;
    CSF_TRY(csf_group_norm(W, wsp(F), 1, 0, Tm, H->h[0], H->w[0], nz, z, d.fuse1_gn, &sc, &sh, &al));
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
      r.cpi = 1; r.out_nstride = (long long)r.Ho * r.Wo;
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
  if (const char* e = getenv("CSF_Z_LDS_MAX")) H->z_lds_max = atoi(e);
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
  // the refresh job table (weight images first, then the small parameter vectors as one-row images)
  std::vector<CsfPrepJobDev> jobs;
  auto push = [&](const CsfPrepArgs& a, int64_t src_off, int64_t dst_off) {
    CsfPrepJobDev j{};
    j.src_off = src_off; j.dst_off = dst_off;
    j.M = a.M; j.Mp = a.Mp; j.Kp = a.Kp; j.ld = a.ld; j.taps = a.taps; j.nseg = a.nseg;
    for (int s = 0; s < CSF_MAX_SEG; ++s) j.seg[s] = a.seg[s];
    const int64_t n = (int64_t)a.Mp * a.Kp;
    j.nblk = (int)std::max<int64_t>(1, std::min<int64_t>(64, (n + 8 * CSN_BLOCK - 1) / (8 * CSN_BLOCK)));
    j.blk0 = H->job_blocks;
    H->job_blocks += j.nblk;
    jobs.push_back(j);
  };
  for (const CsfPrepJob& j : H->prep) push(j.a, j.src_off, j.dst_off);
  for (const CsfCopyJob& c : H->copies) {
    CsfPrepArgs a{};
    a.M = 1; a.Mp = 1; a.Kp = c.n; a.ld = c.n; a.taps = 0; a.nseg = 1;
    a.seg[0] = CsfPrepSeg{0, c.n, 0};
    push(a, c.src, c.dst);
  }
  H->njobs = (int)jobs.size();
  e = hipMalloc(reinterpret_cast<void**>(&H->jobs_dev), jobs.size() * sizeof(CsfPrepJobDev));
  if (e == hipSuccess) e = hipMemcpy(H->jobs_dev, jobs.data(), jobs.size() * sizeof(CsfPrepJobDev), hipMemcpyHostToDevice);
  if (e != hipSuccess) { csf_head_destroy(H); return hip_fail(e, "refresh job table"); }
  *out = H;
  return CSN_OK;
}

void csf_head_destroy(csf_head* H) {
  if (!H) return;
  if (H->packed) (void)hipFree(H->packed);
  if (H->jobs_dev) (void)hipFree(H->jobs_dev);
  delete H;
}

size_t csf_head_workspace_bytes(const csf_head* H) { return H ? H->ws_bytes : 0; }
int64_t csf_head_macs(const csf_head* H) { return H ? H->macs : 0; }

int csf_head_refresh_params(csf_head* H, const float* arena, int64_t arena_floats, void* stream) {
  if (!H || !arena) FAIL(CSN_E_INVALID, "null argument");
  for (const CsfPrepJob& j : H->prep)
    if (j.src_off < 0 || j.src_off + (int64_t)j.a.M * j.a.ld > arena_floats) FAIL(CSN_E_INVALID, "weight offset outside the arena");
  for (const CsfCopyJob& c : H->copies)
    if (c.src < 0 || c.src + c.n > arena_floats) FAIL(CSN_E_INVALID, "parameter offset outside the arena");
  LAUNCH_TRY(csf_launch_prep_all(H->jobs_dev, H->njobs, H->job_blocks, arena, H->packed, stream));
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

int csf_bn_act(float* x, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
               const float* residual, int32_t batch, int32_t channels, int32_t hw, int32_t relu, void* stream) {
  if (!x || !gamma || !beta || !mean || !var || batch < 1 || channels < 1 || hw < 1) FAIL(CSN_E_INVALID, "csf_bn_act");
  CsfBnActArgs a{};
  a.x = x; a.res = residual; a.gamma = gamma; a.beta = beta; a.mean = mean; a.var = var; a.eps = eps;
  a.C = channels; a.HW = hw; a.relu = relu;
  LAUNCH_TRY(csf_launch_bn_act(a, batch * channels, stream));
  return CSN_OK;
}

}  // extern "C"
