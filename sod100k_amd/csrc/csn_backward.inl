// csn_backward.inl -- backward planning + sequencing of the train step (included by csn_plan.hip).
//
// Reference: autograd through CSNet.forward (csnet.py:365-387) as driven by train.py:203-216
//   loss = BCEWithLogits(model(x), t) + FLOPS.WEIGHT * model.get_flops();  loss.backward()
// For every unit, in reverse order:
//   1. BN(train) + PReLU backward per output branch (k_train.hip): dy (from up to two consumers) and the saved
//      raw conv output z -> dz (written over z), d gamma / d beta / d alpha (+ the penalty's d/d gamma);
//   2. weight gradient: one k_wgrad.hip launch per forward contraction pass (same gathered vector as the forward);
//   3. input gradient.  gOctConv (csnet.py:664-726) for input branch i:
//        dx_i = sum_{j == i} W_ij^T dz_j                              own resolution
//             + sum_{j <  i} W_ij^T adjoint_up(dz_j)                  x_i was bilinearly upsampled into y_j; the
//                                                                    1x1 (3x3: tap-flipped) transpose commutes
//                                                                    with the linear resampling
//             + sum_{j >  i} maxpool_backward(W_ij^T dz_j)            x_i was max-pooled into y_j
//      The first two are ONE forward-kernel launch (goct_pw_kernel with transposed, tap-flipped weight blocks
//      and an identity epilogue), the third a launch at the low resolution into a temporary followed by the
//      arg-max routing kernel.  stride 2: the gradient of the pooled copy goes through the avg-pool adjoint.
//      Depthwise: the forward depthwise kernel with flipped taps.  MSBlock: dilated taps, flipped, per dilation.

namespace {

struct AdjPlan {      // adjoint-upsampled dz of output branch j at the resolution of input branch i
  int j = 0, i = 0, f = 2, C = 0, lvl = 0;
  int64_t off = 0;    // bytes inside the unit scratch
};

struct DataLaunch {
  PwLaunchPlan L;
  int i = 0;          // input branch the gradient belongs to
  bool to_tmp = false;
  int pool_f = 0;     // to_tmp: max-pool factor of the routing kernel that follows
  int lvl_lo = 0;     // to_tmp: level of the temporary
  // round 6 (`route`): the branch's pooled term of factor 2 is contracted into the temporary FIRST (to_tmp chunks with route set: no
  // routing kernel behind them) and the direct launch's chunks route it in their epilogue (PwqArgs::route_x); a direct chunk that
  // cannot (not on pwq_kernel, width not a multiple of 4, CSN_POOL_ROUTE=0) leaves it to the routing kernel after the LAST direct chunk
  bool route = false;
  bool route_last = false;
  bool pooled = false;   // a chunk of a pooled term's launch (to_tmp itself marks the LAST chunk only: the routing kernel follows it)
};

struct WgRowSrc {     // `n` consecutive channels, starting at c0, of the tensor (kind, idx) with ctot channels
  int kind = SRC_DZ, idx = 0, c0 = 0, ctot = 0, n = 0;
};

struct WgPlan {
  PwLaunchPlan L;     // one pass, r = 0, L.lvl = absolute level (weight image unused)
  std::vector<WgRowSrc> rows;   // the dz rows of the pass, source after source (sum of n == nrows of the pass)
  std::vector<WgBlock> blocks;
  void one_source(int kind, int idx, int c0, int ctot) {
    WgRowSrc r; r.kind = kind; r.idx = idx; r.c0 = c0; r.ctot = ctot; r.n = L.passes[0].nrows;
    rows.assign(1, r);
  }
};

struct UnitBwd {
  std::vector<AdjPlan> adj;
  std::vector<DataLaunch> data;
  std::vector<WgPlan> wg;
  bool need_dx[3] = {false, false, false};
  bool zero_dx[3] = {false, false, false};   // no own-resolution term: clear before the pooled terms are added
  int64_t dxp_off[3] = {-1, -1, -1};         // stride 2: gradient of the pooled copies (unit scratch)
  int64_t tmp_off = -1;
  int64_t dwf_w[3] = {-1, -1, -1};           // depthwise: tap-flipped x100 weights (packed)
  int64_t scratch = 0;                       // bytes of unit scratch used
  int64_t msdx_w[5] = {-1, -1, -1, -1, -1};  // MSBlock: CSN_PREP_MSDX weight images of ms_dx_kernel, msdx_ng > 0
  int msdx_ng = 0;
  int adj_fused[3] = {-1, -1, -1};           // output branch j: index of the f = 2 AdjPlan its BatchNorm-backward apply pass also
                                             // produces (bn_bwd_apply_adj2_kernel; dz then goes to the branch's activation buffer)
};

// one DataLaunch per row chunk of `L`; the routing kernel (to_tmp) follows the last chunk only
void push_data(UnitBwd& ub, const PwLaunchPlan& L, const DataLaunch& proto) {
  std::vector<PwLaunchPlan> parts;
  add_launch(parts, L);
  for (size_t k = 0; k < parts.size(); ++k) {
    DataLaunch dl = proto;
    dl.L = parts[k];
    dl.to_tmp = proto.to_tmp && k + 1 == parts.size();
    dl.pooled = proto.to_tmp;
    dl.route_last = proto.route && !proto.to_tmp && k + 1 == parts.size();
    ub.data.push_back(dl);
  }
}

int64_t bw_alloc(UnitBwd& ub, int64_t bytes) {
  const int64_t off = ub.scratch;
  ub.scratch += align_up(bytes, 256);
  return off;
}

void ident_epi(const csn_plan& P, PwPassPlan& ps) { ps.epi = P.ident; }

WgPlan make_wg(const PwLaunchPlan& L, const PwPassPlan& pp) {
  WgPlan w;
  w.L.lvl = L.lvl + pp.r;
  PwPassPlan q = pp;
  q.r = 0;
  w.L.passes.push_back(q);
  for (const WBlock& b : pp.wb) {
    if (b.eye) continue;
    WgBlock g;
    g.dst = b.src; g.ld = b.ld; g.ncol = b.ncol; g.col = b.col; g.scale = b.scale; g.tk = b.tk;
    w.blocks.push_back(g);
  }
  return w;
}

// an input branch gets a gradient when it is an activation of the net -- or the image batch itself under CSN_OPT_INPUT_GRAD
// (autograd's x.grad: csn_train_act_info(0).grad_offset_bytes[0] after csn_backward)
static inline bool wants_dx(const csn_plan& P, int act) { return act > 0 || (act == 0 && P.input_grad); }

int plan_goct_bwd(Builder& bl, UnitPlan& u, UnitBwd& ub) {
  csn_plan& P = bl.P;
  const csn_unit_desc& d = u.d;
  const int kk = d.ksize * d.ksize;
  const int base = u.base_lvl;
  int cin_tot = 0, cout_tot = 0, ci_off[4] = {0}, co_off[4] = {0};
  for (int i = 0; i < d.n_in; ++i) { ci_off[i] = cin_tot; cin_tot += d.cin[i]; }
  for (int j = 0; j < d.n_out; ++j) { co_off[j] = cout_tot; cout_tot += d.cout[j]; }
  const int ld = cin_tot * kk;
  for (int i = 0; i < d.n_in; ++i) ub.need_dx[i] = d.cin[i] > 0 && wants_dx(P, d.in_act[i]);
  auto adj_index = [&](int j, int i) {
    for (size_t k = 0; k < ub.adj.size(); ++k)
      if (ub.adj[k].j == j && ub.adj[k].i == i) return (int)k;
    AdjPlan a;
    a.j = j; a.i = i; a.f = 1 << (i - j); a.C = d.cout[j]; a.lvl = base + i;
    a.off = bw_alloc(ub, bl.act_bytes(a.C, a.lvl));
    ub.adj.push_back(a);
    return (int)ub.adj.size() - 1;
  };
  // ---- weight gradients.  Multi-branch 1x1 units: the low -> high blocks move the bilinear adjoint onto dz (the same
  // adjoint-upsampled dz_j the input gradient needs) and are contracted at the LOW resolution,
  //     dW[co_j][ci_i] = sum_{p at res i} adjoint_up(dz_j)[co_j][p] * x_i[ci_i][p]          (i > j)
  // so no weight-gradient launch gathers bilinear taps (4 loads per channel and pixel at the finest resolution) and the
  // pass at resolution r is  [adj(dz_0); ..; adj(dz_{r-1}); dz_r] x x_r  (rows = consecutive rows of W, columns = block r)
  // plus  dz_r x [pool(x_0); ..; pool(x_{r-1})]  for the high -> low blocks.
  // (an output branch wider than a launch holds -- the un-pruned x2 net's 132 / 144-channel heads -- is a pass of its own with ONE row
  // source, which run_wgrad takes in row chunks; until round 6 such units kept the forward-shaped passes with bilinear gathers)
  const bool regroup = d.ksize == 1 && !u.std_conv && (d.n_in > 1 || d.n_out > 1) && std::getenv("CSN_WGRAD_REGROUP") == nullptr;
  if (regroup) {
    for (int r = 0; r < d.n_in; ++r) {
      if (d.cin[r] == 0) continue;
      // rows = the output branches j <= r, as many per launch as the kernel holds (80 rows)
      int j = 0;
      while (j <= r && j < d.n_out) {
        WgPlan w;
        w.L.lvl = base + r;
        PwPassPlan q;
        q.r = 0; q.nsrc = 1; q.src_kind[0] = SRC_IN; q.src_branch[0] = r; q.src_C[0] = d.cin[r]; q.src_mode[0] = PW_OWN;
        q.K = d.cin[r];
        int first = -1;
        for (; j <= r && j < d.n_out; ++j) {
          if (d.cout[j] == 0) continue;
          if (r - j > 2) FAIL(CSN_E_UNSUPPORTED, "bilinear factor > 4");
          // <= 48 rows (and K <= 64) keeps the launch on the wave-private kernel (no block barrier per pixel group)
          const int row_cap = d.cin[r] <= 64 ? 48 : WG_MAX_ROWS;
          if ((q.nrows > 0 && q.nrows + d.cout[j] > row_cap) || w.rows.size() == 3) break;
          if (first < 0) first = j;
          WgRowSrc rs;
          rs.c0 = 0; rs.ctot = d.cout[j]; rs.n = d.cout[j];
          if (j == r) { rs.kind = SRC_DZ; rs.idx = j; }
          else { rs.kind = SRC_ADJ; rs.idx = adj_index(j, r); }
          w.rows.push_back(rs);
          q.nrows += d.cout[j];
        }
        if (q.nrows == 0) continue;
        w.L.passes.push_back(q);
        WgBlock g;
        g.dst = d.w_off[0] + (int64_t)co_off[first] * cin_tot + ci_off[r]; g.ld = cin_tot; g.ncol = d.cin[r]; g.col = 0;
        g.scale = 1.f; g.tk = 0;
        w.blocks.push_back(g);
        ub.wg.push_back(w);
      }
    }
    for (int j = 1; j < d.n_out; ++j) {
      if (d.cout[j] == 0) continue;
      WgPlan w;
      w.L.lvl = base + j;
      PwPassPlan q;
      q.r = 0; q.nrows = d.cout[j];
      for (int i = 0; i < j && i < d.n_in; ++i) {
        if (d.cin[i] == 0) continue;
        if (j - i > 2) FAIL(CSN_E_UNSUPPORTED, "max-pool factor");
        const int s = q.nsrc++;
        q.src_kind[s] = SRC_IN; q.src_branch[s] = i; q.src_C[s] = d.cin[i]; q.src_mode[s] = j - i == 1 ? PW_POOL2 : PW_POOL4;
        WgBlock g;
        g.dst = d.w_off[0] + (int64_t)co_off[j] * cin_tot + ci_off[i]; g.ld = cin_tot; g.ncol = d.cin[i]; g.col = q.K;
        g.scale = 1.f; g.tk = 0;
        w.blocks.push_back(g);
        q.K += d.cin[i];
      }
      if (q.nsrc == 0) continue;
      w.L.passes.push_back(q);
      w.one_source(SRC_DZ, j, 0, d.cout[j]);
      ub.wg.push_back(w);
    }
  }
  // ... everything else: one launch per forward pass (the z launch pairs with the adjoint-upsampled dz_0)
  if (!regroup)
  for (const PwLaunchPlan& L : u.pwl)
    for (const PwPassPlan& pp : L.passes) {
      PwPassPlan q = pp;
      if (q.nsrc > 0 && q.src_kind[q.nsrc - 1] == SRC_Z) {   // identity columns carry no parameter
        q.nsrc -= 1;
        q.K -= u.z_C;
      }
      if (q.nsrc == 0 || q.K == 0) continue;
      WgPlan w = make_wg(L, q);
      if (pp.out_kind == OUT_Z) w.one_source(SRC_ADJ, adj_index(0, 1), pp.out_c0, u.z_C);
      else w.one_source(SRC_DZ, pp.out_branch, pp.out_c0, d.cout[pp.out_branch]);
      ub.wg.push_back(w);
    }
  // ---- input gradients
  const bool std_s2 = u.std_conv && d.stride == 2;   // real stride: the input gradient is the zero-stuffed adjoint
  const int mode = d.ksize == 3 ? (std_s2 ? PW_TAPS_UPS2 : PW_TAPS) : PW_OWN;
  int64_t tmp_bytes = 0;
  for (int i = 0; i < d.n_in; ++i) {
    if (!ub.need_dx[i]) continue;
    if (d.stride == 2 && !std_s2) ub.dxp_off[i] = bw_alloc(ub, bl.act_bytes(d.cin[i], base + i));
    auto wblk = [&](int j) {
      WBlock w;
      w.src = d.w_off[0] + ((int64_t)co_off[j] * cin_tot + ci_off[i]) * kk;
      w.ld = ld; w.ncol = d.cout[j] * kk; w.tk = kk;
      if (u.std_conv) w.scale = 100.f;
      return w;
    };
    PwLaunchPlan L;
    L.lvl = base + i - (std_s2 ? 1 : 0);
    PwPassPlan ps;
    ps.r = 0; ps.nrows = d.cin[i]; ps.out_kind = OUT_DX; ps.out_branch = i; ps.out_ctot = d.cin[i];
    ident_epi(P, ps);
    auto add = [&](PwPassPlan& q, int kind, int idx, int C) {
      const int s = q.nsrc++;
      q.src_kind[s] = kind; q.src_branch[s] = idx; q.src_C[s] = C; q.src_mode[s] = mode;
      return s;
    };
    for (int j = 0; j <= i && j < d.n_out; ++j) {
      if (d.cout[j] == 0) continue;
      if (ps.nsrc >= 3) FAIL(CSN_E_UNSUPPORTED, "backward: too many sources");
      WBlock w = wblk(j);
      w.col = ps.K;
      if (j == i) add(ps, SRC_DZ, j, d.cout[j]);
      else add(ps, SRC_ADJ, adj_index(j, i), d.cout[j]);
      ps.wb.push_back(w);
      ps.K += d.cout[j] * kk;
    }
    // one pooled term with factor 2 behind a direct launch of a 1x1 unit: temporary first, routed by the direct launch (DataLaunch::route)
    // (a second pooled term -- factor 4, CSFHead.fuse's first input -- keeps its routing kernel, behind the direct launch)
    const bool route = P.pool_route && ps.nsrc > 0 && i + 1 < d.n_out && d.cout[i + 1] > 0 && (mode == PW_OWN || mode == PW_TAPS);
    auto push_direct = [&]() {
      L.passes.push_back(ps);
      DataLaunch dl;
      dl.i = i; dl.route = route; dl.pool_f = 2; dl.lvl_lo = base + i + 1;
      push_data(ub, L, dl);
    };
    if (ps.nsrc > 0) {
      if (!route) push_direct();
    } else {
      ub.zero_dx[i] = true;
    }
    for (int j = i + 1; j < d.n_out; ++j) {   // x_i was max-pooled into y_j
      if (d.cout[j] == 0) continue;
      PwLaunchPlan Lp;
      Lp.lvl = base + j;
      PwPassPlan pq;
      pq.r = 0; pq.nrows = d.cin[i]; pq.out_kind = OUT_TMP; pq.out_ctot = d.cin[i];
      ident_epi(P, pq);
      add(pq, SRC_DZ, j, d.cout[j]);
      WBlock w = wblk(j);
      w.col = 0;
      pq.wb.push_back(w);
      pq.K = d.cout[j] * kk;
      Lp.passes.push_back(pq);
      DataLaunch dl;
      dl.i = i; dl.to_tmp = true; dl.pool_f = 1 << (j - i); dl.lvl_lo = base + j; dl.route = route && j == i + 1;
      push_data(ub, Lp, dl);
      tmp_bytes = std::max(tmp_bytes, bl.act_bytes(d.cin[i], base + j));
      if (route && j == i + 1) push_direct();
    }
  }
  if (tmp_bytes > 0) ub.tmp_off = bw_alloc(ub, tmp_bytes);
  for (DataLaunch& dl : ub.data) {
    const int st = finish_launch(bl, dl.L);
    if (st != CSN_OK) return st;
  }
  // the x2 adjoint upsampling of dz_j inside the BatchNorm backward's apply pass of branch j (k_train.hip)
  for (size_t k = 0; k < ub.adj.size(); ++k) {
    const AdjPlan& ap = ub.adj[k];
    const int a = d.out_act[ap.j];
    if (ap.f != 2 || a <= 0 || P.acts[a].ws_off < 0) continue;
    const int lvl = base + ap.j;
    const int Hh = P.H >> lvl, Wh = P.W >> lvl;
    if (csn_bn_bwd_adj2_ok((int64_t)Hh * Wh, Wh)) ub.adj_fused[ap.j] = (int)k;
  }
  return CSN_OK;
}

int plan_dw_bwd(Builder& bl, UnitPlan& u, UnitBwd& ub) {
  const csn_unit_desc& d = u.d;
  for (int k = 0; k < d.n_in; ++k) {
    if (d.cout[k] == 0) continue;
    ub.need_dx[k] = wants_dx(bl.P, d.in_act[k]);
    ub.dwf_w[k] = bl.alloc_packed((int64_t)d.cout[k] * 9);
    bl.job(CSN_PREP_FLIP9, d.cout[k] * 9, ub.dwf_w[k], d.w_off[k], -1, -1, -1, 100.0f);
  }
  return CSN_OK;
}

int plan_ms_bwd(Builder& bl, UnitPlan& u, UnitBwd& ub) {
  csn_plan& P = bl.P;
  const csn_unit_desc& d = u.d;
  const int cin = d.cin[0], cout = d.cout[0];
  ub.need_dx[0] = wants_dx(P, d.in_act[0]);
  int cobase[CSN_NDIL], base = 0;
  for (int k = 0; k < CSN_NDIL; ++k) { cobase[k] = base; base += d.dil_ch[k]; }
  // weight gradient with the roles swapped: rows = the cin input channels (A = x), gathered = the dilated taps of the
  // dz slices -- sum_d 9*dch_d gathered entries per pixel instead of 9*cin per dilation (5-10x fewer loads):
  //   dW_d[co][ci][t] = 100 * sum_p' x[ci][p'] * dz[co][p' + dil*off(8 - t)]
  for (int k0 = 0; k0 < CSN_NDIL;) {
    PwLaunchPlan L;
    L.lvl = u.base_lvl;
    PwPassPlan ps;
    ps.r = 0; ps.nrows = cin;
    for (; k0 < CSN_NDIL && ps.nsrc < 3; ++k0) {
      if (d.dil_ch[k0] == 0) continue;
      const int s = ps.nsrc++;
      ps.src_kind[s] = SRC_DZ; ps.src_branch[s] = 0; ps.src_C[s] = d.dil_ch[k0]; ps.src_mode[s] = PW_TAPS;
      ps.src_dil[s] = 1 << k0; ps.src_c0[s] = cobase[k0]; ps.src_ctot[s] = cout;
      WBlock w; w.src = d.w_off[k0]; w.ld = cin * 9; w.ncol = d.dil_ch[k0] * 9; w.col = ps.K; w.scale = 100.f; w.tk = 9;
      ps.wb.push_back(w);
      ps.K += d.dil_ch[k0] * 9;
    }
    if (ps.nsrc == 0) break;
    WgPlan wg = make_wg(L, ps);
    wg.one_source(SRC_IN, 0, 0, cin);
    ub.wg.push_back(wg);
  }
  if (!ub.need_dx[0]) return CSN_OK;
  // backward data: ms_dx_kernel (one launch, every dz tap loaded once per pixel); CSN_MS_DX=0: the generic tap kernel below
  if (P.ms_dx && cin <= 56) {   // (round 6: up to seven groups of eight sums -- the un-pruned x2 net's 53-channel MSBlocks)
    ub.msdx_ng = (cin + 7) / 8;
    for (int k = 0; k < CSN_NDIL; ++k) {
      if (d.dil_ch[k] == 0) continue;
      const int ncop = (d.dil_ch[k] + 1) & ~1;
      ub.msdx_w[k] = bl.alloc_packed((int64_t)ncop * 9 * ub.msdx_ng * 8);
      bl.job(CSN_PREP_MSDX, d.dil_ch[k], ub.msdx_w[k], d.w_off[k], -1, -1, -1, 100.0f, cin, 0, ub.msdx_ng * 8, 0);
    }
    return CSN_OK;
  }
  int k = 0;
  bool first = true;
  while (k < CSN_NDIL) {
    PwLaunchPlan L;
    L.lvl = u.base_lvl;
    PwPassPlan ps;
    ps.r = 0; ps.nrows = cin; ps.out_kind = OUT_DX; ps.out_branch = 0; ps.out_ctot = cin;
    ident_epi(P, ps);
    const int room = first ? 3 : 2;
    for (; k < CSN_NDIL && ps.nsrc < room; ++k) {
      if (d.dil_ch[k] == 0) continue;
      const int s = ps.nsrc++;
      ps.src_kind[s] = SRC_DZ; ps.src_branch[s] = 0; ps.src_C[s] = d.dil_ch[k]; ps.src_mode[s] = PW_TAPS;
      ps.src_dil[s] = 1 << k; ps.src_c0[s] = cobase[k]; ps.src_ctot[s] = cout;
      WBlock w; w.src = d.w_off[k]; w.ld = cin * 9; w.ncol = d.dil_ch[k] * 9; w.col = ps.K; w.scale = 100.f; w.tk = 9;
      ps.wb.push_back(w);
      ps.K += d.dil_ch[k] * 9;
    }
    if (ps.nsrc == 0) break;
    if (!first) {   // accumulate onto the dilations of the previous launch through an identity block
      const int s = ps.nsrc++;
      ps.src_kind[s] = SRC_DX; ps.src_branch[s] = 0; ps.src_C[s] = cin; ps.src_mode[s] = PW_OWN;
      WBlock w; w.eye = 1; w.ncol = cin; w.col = ps.K;
      ps.wb.push_back(w);
      ps.K += cin;
    }
    L.passes.push_back(ps);
    DataLaunch dl;
    dl.i = 0;
    push_data(ub, L, dl);
    first = false;
    bool more = false;
    for (int q = k; q < CSN_NDIL; ++q) more = more || d.dil_ch[q] > 0;
    if (!more) break;
  }
  for (DataLaunch& dl : ub.data) {
    const int st = finish_launch(bl, dl.L);
    if (st != CSN_OK) return st;
  }
  return CSN_OK;
}

int plan_cls_bwd(Builder& bl, UnitPlan& u, UnitBwd& ub) {
  csn_plan& P = bl.P;
  const csn_unit_desc& d = u.d;
  ub.need_dx[0] = wants_dx(P, d.in_act[0]);
  {
    // dW[0][ci] = sum_p dlogit[p] * x[ci][p]: ONE row against 79 gathered channels would leave 15 of the 16 MFMA rows and
    // the wave-private kernel (K <= 64) unused -- swap the roles: rows = the input channels, gathered = the logit gradient
    WgPlan wg;
    wg.L.lvl = u.pwl[0].lvl;
    PwPassPlan q;
    q.r = 0; q.nsrc = 1; q.src_kind[0] = SRC_DZ; q.src_branch[0] = 0; q.src_C[0] = 1; q.src_mode[0] = PW_OWN;
    q.K = 1; q.nrows = d.cin[0];
    wg.L.passes.push_back(q);
    WgBlock g;
    g.dst = d.w_off[0]; g.ld = d.cin[0]; g.ncol = 1; g.col = 0; g.scale = 1.f; g.tk = 1;
    wg.blocks.push_back(g);
    wg.one_source(SRC_IN, 0, 0, d.cin[0]);
    ub.wg.push_back(wg);
  }
  if (!ub.need_dx[0]) return CSN_OK;
  PwLaunchPlan L;
  L.lvl = 1;
  PwPassPlan ps;
  ps.r = 0; ps.nsrc = 1; ps.src_kind[0] = SRC_DZ; ps.src_branch[0] = 0; ps.src_C[0] = 1; ps.src_mode[0] = PW_OWN;
  ps.K = 1; ps.nrows = d.cin[0]; ps.out_kind = OUT_DX; ps.out_branch = 0; ps.out_ctot = d.cin[0];
  ident_epi(P, ps);
  WBlock w; w.src = d.w_off[0]; w.ld = d.cin[0]; w.ncol = 1; w.col = 0; w.tk = 1;
  ps.wb.push_back(w);
  L.passes.push_back(ps);
  DataLaunch dl;
  dl.i = 0;
  push_data(ub, L, dl);
  for (DataLaunch& one : ub.data) {
    const int st = finish_launch(bl, one.L);
    if (st != CSN_OK) return st;
  }
  return CSN_OK;
}

// ---------------------------------------------------------------------------------------------- execution
#define CSN_WG_REGIONS 8   // partial-dW regions: that many weight-gradient passes are reduced by ONE wgrad_reduce launch
struct BwdDefer {          // launches of csn_backward that only feed the optimizer: collected, issued in batches
  std::vector<WgReduceArgs> wgred;   // pending reductions, one partial region each
  std::vector<DwFinJob> dwfin;       // depthwise finalise jobs (own partial regions: UnitPlan::dwwg_off), flushed at the end
  int next_region = 0;
};
struct BwdCtx {
  const Ctx& c;
  const float* arena;
  float* grad;
  const float* flop_w;
  float pen_scale;
  BwdDefer* defer = nullptr;         // null: every reduction right behind its pass (side-lane mode)
};

int flush_wgred(const BwdCtx& b) {
  if (!b.defer || b.defer->wgred.empty()) return CSN_OK;
  LAUNCH_TRY(csn_launch_wgrad_reduce_batch(b.defer->wgred.data(), (int)b.defer->wgred.size(), b.c.stream));
  b.defer->wgred.clear();
  return CSN_OK;
}

float* grad_buf(const Ctx& c, int act, int slot) {
  return reinterpret_cast<float*>(c.ws + c.P.tg_off[act][slot]);
}

int run_wgrad(const BwdCtx& b, const WgPlan& w, const PwBind& bd) {
  const csn_plan& P = b.c.P;
  const PwPassPlan& pp = w.L.passes[0];
  WgArgs a;
  fill_pass(b.c, w.L, pp, bd, a.ps);
  a.Hr = P.H >> w.L.lvl; a.Wr = P.W >> w.L.lvl; a.B = P.S;
  const int64_t hw = (int64_t)a.Hr * a.Wr;
  auto row_base = [&](const WgRowSrc& r) {
    return r.kind == SRC_ADJ ? bd.adj[r.idx] : (r.kind == SRC_IN ? bd.in[r.idx] : bd.dz[r.idx]);
  };
  if (w.rows.empty() || w.rows.size() > 3 || (w.rows.size() > 1 && pp.nrows > WG_MAX_ROWS)) return CSN_E_INVALID;
  a.nrs = (int)w.rows.size();
  for (int q = 0; q < 3; ++q) { a.rs[q].ptr = nullptr; a.rs[q].ctot = 0; a.rs[q].n = 0; }
  for (int q = 0; q < a.nrs; ++q) {
    a.rs[q].ptr = b.c.eo(row_base(w.rows[q]), (int64_t)w.rows[q].c0 * hw);
    a.rs[q].ctot = w.rows[q].ctot; a.rs[q].n = w.rows[q].n;
  }
  // a 2x2 max-pooled tap slice whose pooled copy exists (c3q_kernel's forward launch read it: u.mp_off, written in csn_forward_train)
  // is read as a plain tap slice of that copy -- the same values, a quarter of the loads
  for (int s2 = 0; s2 < a.ps.nsrc; ++s2)
    if (a.ps.src[s2].mode == PW_POOL2_TAPS && pp.src_kind[s2] == SRC_IN && bd.mp[pp.src_branch[s2]] != nullptr) {
      a.ps.src[s2].ptr = b.c.eo(bd.mp[pp.src_branch[s2]], (int64_t)pp.src_c0[s2] * hw);
      a.ps.src[s2].mode = PW_TAPS;
    }
  a.gpp = (int)((hw + 63) / 64);
  a.ngroups = a.gpp * P.S;
  a.k16 = (pp.K + 15) & ~15;
  a.partial = reinterpret_cast<float*>(b.c.ws + (b.c.side ? P.wg2_off : P.wg_off));
  a.a16 = b.c.a16 ? 1 : 0; a.pad = 0;
  const bool defer = b.defer != nullptr && !b.c.side;
  // the kernel holds at most 80 output channels (5 MFMA row tiles) per launch: wider passes go in row chunks
  // ... and passes with K <= 64 in chunks of 48 rows, which keeps them on the wave-private kernel
  int row_chunk = (a.nrs == 1 && a.k16 <= 64 && pp.nrows > 48) ? 48 : WG_MAX_ROWS;
  // bf16 storage, dilated tap passes (MSBlock): wgrad_bf16_c3_kernel's dilated forms hold three row tiles -- the un-pruned net's
  // 53-channel blocks go in chunks of 48 rows instead of falling back to the generic kernel
  if (a.a16 && a.nrs == 1 && pp.nrows > 48) {
    WgArgs t = a;
    t.ps.nrows = std::min(pp.nrows, row_chunk); t.rows16 = (t.ps.nrows + 15) & ~15;
    WgArgs t48 = a;
    t48.ps.nrows = 48; t48.rows16 = 48;
    if (!csn_wgrad_bf3_eligible(t) && csn_wgrad_bf3_eligible(t48)) row_chunk = 48;
  }
  // Column pieces (round 6).  bf16 storage, 1x1 passes beyond wgrad_bf16_kernel's 4 x 4-tile limit (the un-pruned x2 net: 80 .. 160 rows
  // against 160 gathered channels) fell back to the generic kernel -- 13.6 of 87 ms per step.  The columns of dW are independent dot
  // products: such a pass goes in row chunks of 64 and pieces of <= 128 channels of ONE source, each a launch of the bf16 kernel with its
  // own partial region; the reduction takes the part of every weight block that the piece covers.
  struct Piece { int s, c0, n, k0; };
  std::vector<Piece> pieces;
  {
    bool own = true, pool = true, flat = true;
    for (int s2 = 0; s2 < a.ps.nsrc; ++s2) {
      own = own && a.ps.src[s2].mode == PW_OWN;
      pool = pool && a.ps.src[s2].mode == PW_POOL2;
    }
    for (size_t q = 0; q < w.blocks.size() && q < 3; ++q) flat = flat && w.blocks[q].tk == 0;
    WgArgs t = a;
    t.ps.nrows = std::min(pp.nrows, row_chunk); t.rows16 = (t.ps.nrows + 15) & ~15;
    if (a.a16 && (own || pool) && flat && w.blocks.size() <= 3 && !csn_wgrad_bf_eligible(t)) {
      // (several row sources -- the regrouped passes, <= 80 rows -- cannot go in row chunks: narrower pieces instead.  The kernel
      // takes at most 4 x 4 tiles of 32 rows / channels and 8 tiles in all)
      const int rc = a.nrs == 1 ? 64 : std::min(pp.nrows, row_chunk);   // (96- and 128-row chunks measured: no difference)
      const int ntr = (rc + 31) / 32;
      const int pw = 32 * std::max(1, std::min(4, 8 / ntr));
      int k0 = 0;
      for (int s2 = 0; s2 < a.ps.nsrc; ++s2) {
        for (int c0 = 0; c0 < a.ps.src[s2].C; c0 += pw) pieces.push_back(Piece{s2, c0, std::min(pw, a.ps.src[s2].C - c0), k0 + c0});
        k0 += a.ps.src[s2].C;
      }
      WgArgs t2 = a;   // would a piece be taken?  (geometry limits of the bf16 kernel: HW % 16, pooled widths ...)
      t2.ps.nrows = std::min(pp.nrows, rc); t2.rows16 = (t2.ps.nrows + 15) & ~15;
      t2.ps.nsrc = 1; t2.ps.src[0] = a.ps.src[pieces[0].s]; t2.ps.src[0].C = t2.ps.src[0].K = pieces[0].n;
      t2.ps.cin = pieces[0].n; t2.ps.cin4 = (pieces[0].n + 3) & ~3; t2.k16 = (pieces[0].n + 15) & ~15;
      if (csn_wgrad_bf_eligible(t2)) row_chunk = a.nrs == 1 ? rc : row_chunk;
      else pieces.clear();
    }
  }
  if (pieces.empty()) pieces.push_back(Piece{-1, 0, pp.K, 0});   // the whole pass
  for (int r0 = 0; r0 < pp.nrows; r0 += row_chunk) {
    const int nr = std::min(row_chunk, pp.nrows - r0);
    for (const Piece& pc : pieces) {
      WgArgs q = a;
      q.ps.nrows = nr;
      if (q.nrs == 1) {   // row chunks of a single source
        q.rs[0].ptr = b.c.eo(row_base(w.rows[0]), (int64_t)(w.rows[0].c0 + r0) * hw);
        q.rs[0].n = nr;
      }
      if (pc.s >= 0) {
        const PwSrc& src = a.ps.src[pc.s];
        q.ps.nsrc = 1;
        q.ps.src[0] = src;
        q.ps.src[0].ptr = b.c.eo(src.ptr, (int64_t)pc.c0 * (src.mode == PW_POOL2 ? 4 * hw : hw));
        q.ps.src[0].C = q.ps.src[0].K = pc.n;
        q.ps.cin = pc.n; q.ps.cin4 = (pc.n + 3) & ~3;
        q.k16 = (pc.n + 15) & ~15;
      }
      q.rows16 = (nr + 15) & ~15;
      q.nblk = csn_wgrad_blocks(q);
      if (defer) {   // this pass's own partial region; the reduction waits for a batch (CSN_WG_REGIONS passes per launch)
        if ((int)b.defer->wgred.size() >= CSN_WG_REGIONS) { const int fs = flush_wgred(b); if (fs != CSN_OK) return fs; }
        q.partial = reinterpret_cast<float*>(b.c.ws + P.wg_off) + (int64_t)(b.defer->next_region++ % CSN_WG_REGIONS) * P.wg_region_floats;
      }
      LAUNCH_TRY(csn_launch_wgrad(q, b.c.stream));
      WgReduceArgs r;
      r.partial = q.partial; r.grad = b.grad;
      r.nblocks = 0;
      for (int k = 0; k < 3; ++k) { r.blk[k].dst = 0; r.blk[k].ld = 0; r.blk[k].ncol = 0; r.blk[k].col = 0; r.blk[k].scale = 0.f; r.blk[k].tk = 0; }
      for (size_t k = 0; k < w.blocks.size() && k < 3; ++k) {
        WgBlock bk = w.blocks[k];
        bk.dst += (int64_t)r0 * (bk.tk > 0 ? bk.tk : bk.ld);
        if (pc.s >= 0) {   // the columns [pc.k0, pc.k0 + pc.n) of the pass that fall into this block
          const int lo = std::max(bk.col, pc.k0), hi = std::min(bk.col + bk.ncol, pc.k0 + pc.n);
          if (hi <= lo) continue;
          bk.dst += lo - bk.col; bk.col = lo - pc.k0; bk.ncol = hi - lo;
        }
        r.blk[r.nblocks++] = bk;
      }
      if (r.nblocks == 0) continue;
      r.nblk = q.nblk; r.nrows = nr; r.K = pc.s >= 0 ? pc.n : pp.K; r.rows16 = q.rows16; r.k16 = q.k16;
      if (defer) b.defer->wgred.push_back(r);
      else LAUNCH_TRY(csn_launch_wgrad_reduce(r, b.c.stream));
    }
  }
  return CSN_OK;
}

// skip_apply: reduce + finalise only (the unit's depthwise backward forms dz on load); keep: the arguments, for that kernel
// adj2: the apply pass also writes the x2 adjoint-upsampled dz there, and dz itself over the branch's (dead) activation
// steps: 0 = reduce + finalise + apply (one branch on its own); 1 = reduce only, the arguments (with the launcher's cpp / nslab)
// go to *keep: the caller finalises all branches of the unit in ONE launch and then applies (csn_launch_bn_bwd_apply_step)
int run_bn_bwd(const BwdCtx& b, int ui, const UnitPlan& u, int j, bool skip_apply = false, BnBwdArgs* keep = nullptr,
               float* adj2 = nullptr, int steps = 0) {
  const csn_plan& P = b.c.P;
  const csn_unit_desc& d = u.d;
  const int act = d.out_act[j];
  const Act& A = P.acts[act];
  BnBwdArgs a;
  a.dyA = grad_buf(b.c, act, 0);
  a.dyB = P.n_cons[act] > 1 ? grad_buf(b.c, act, 1) : nullptr;
  a.z = reinterpret_cast<float*>(b.c.ws + P.tz_off[act]);
  a.scale = P.packed + u.out_epi[j].scale; a.shift = P.packed + u.out_epi[j].shift; a.alpha = P.packed + u.out_epi[j].alpha;
  a.mean = P.packed + u.tr_mean[j]; a.invstd = P.packed + u.tr_invstd[j];
  // (a third of the table per output branch: the reduce passes of a unit's branches all run before their common finalise launch)
  a.partial = reinterpret_cast<double*>(b.c.ws + P.red_off) + (int64_t)j * P.red_maxc * CSN_BN_NSLAB * 3;
  a.m1m2 = P.packed + u.tr_m1m2[j];
  a.arena = b.arena; a.grad = b.grad;
  a.gapabs = u.gap_off[j] >= 0 ? reinterpret_cast<const float*>(b.c.ws + u.gap_off[j]) : nullptr;
  a.off_weight = d.bn[j].weight; a.off_bias = d.bn[j].bias; a.off_prelu = d.bn[j].prelu;
  a.HW = (int64_t)(P.H >> A.lvl) * (P.W >> A.lvl);
  a.S = P.S; a.C = d.cout[j];
  a.flop_w = b.flop_w[ui * CSN_MAX_BRANCH + j];
  a.pen_scale = b.pen_scale;
  a.a16 = b.c.a16 ? 1 : 0;
  a.skip_apply = skip_apply ? 1 : 0;
  a.nslab_in = 0;
  if (adj2) { a.adj2 = adj2; a.dz_out = b.c.act_y(act); a.W = P.W >> A.lvl; }
  if (!P.virt_cons.empty() && P.virt_cons[act] >= 0) {   // the only consumer's backward kernel has left the partial sums
    const UnitPlan& cu = P.units[P.virt_cons[act]];
    for (int i = 0; i < cu.d.n_in; ++i)
      if (cu.d.cout[i] > 0 && cu.d.in_act[i] == act) {
        a.partial = reinterpret_cast<double*>(b.c.ws + cu.bnred_off[i]);
        a.nslab_in = dw_stats_slabs(P, A.lvl);
      }
  }
  if (steps == 1) {
    LAUNCH_TRY(csn_launch_bn_bwd_reduce(a, b.c.stream));
  } else {
    LAUNCH_TRY(csn_launch_bn_bwd(a, b.c.stream));
  }
  if (keep) *keep = a;
  return CSN_OK;
}

// depthwise unit, branch k: input and weight gradient in ONE pass over dz and x (dw3x3_bwd_kernel) -- when both are wanted and
// the per-(image, tile) partials fit the reduction table; not on the weight-gradient side lane (the fused kernel also writes the
// input gradient that the NEXT unit's backward on the caller's stream reads)
// virt: the unit's input was never stored (csn_plan::virt_cons) -- then this IS the path, on the caller's stream
int dw_fused_slabs(const csn_plan& P, const UnitBwd& ub, int k, int lvl, bool side, bool virt) {
  if (virt) return dw_stats_slabs(P, lvl);
  return (ub.need_dx[k] && !side && !std::getenv("CSN_DW_BWD_SPLIT")) ? dw_stats_slabs(P, lvl) : 0;
}
inline bool dw_in_virtual(const csn_plan& P, int ui, int k) {
  const int a = P.units[ui].d.in_act[k];
  return !P.virt_cons.empty() && a > 0 && P.virt_cons[a] == ui;
}

int run_unit_bwd(const BwdCtx& b, int ui, const float* dy) {
  const Ctx& c = b.c;
  csn_plan& P = c.P;
  const UnitPlan& u = P.units[ui];
  const UnitBwd& ub = P.bwd[ui];
  const csn_unit_desc& d = u.d;
  const int S = P.S;
  char* scratch = c.ws + P.scratch_off;
  PwBind bd;
  BnBwdArgs bnargs[CSN_MAX_BRANCH];
  bool bn_fused[CSN_MAX_BRANCH] = {false, false, false};
  if (d.kind == CSN_UNIT_CLS) {
    float* dlh = reinterpret_cast<float*>(c.ws + u.logits_off);   // gradient of the half-resolution logits
    AdjUpArgs ua;
    ua.in = dy; ua.out = dlh; ua.planes = S; ua.Hl = P.H >> 1; ua.Wl = P.W >> 1; ua.f = 2;
    ua.in16 = 0; ua.out16 = c.a16 ? 1 : 0;   // dy is the caller's float gradient of the logits
    LAUNCH_TRY(csn_launch_adjup(ua, c.stream));
    LAUNCH_TRY(csn_launch_sum_to_grad(dlh, (int64_t)S * ua.Hl * ua.Wl, b.grad + d.bias_off,
                                      reinterpret_cast<double*>(c.ws + P.red_off), c.a16 ? 1 : 0, c.stream));
    bd.in[0] = c.act_in(d.in_act[0]);
    bd.dz[0] = dlh;
  } else {
    for (int j = 0; j < d.n_out; ++j) {
      if (d.cout[j] == 0) continue;
      // depthwise units on the one-pass backward kernel: that kernel forms dz on load, the apply pass is skipped (round 3)
      bool fuse_apply = false;
      if (d.kind == CSN_UNIT_DW && P.bn_bwd_fuse)
        fuse_apply = dw_fused_slabs(P, ub, j, P.acts[d.in_act[j]].lvl, c.lanes, dw_in_virtual(P, ui, j)) > 0;
      // (not with the weight-gradient side lane, CSN_OPT_OVERLAP = 2: the fused pass writes dz over the branch's activation, which
      // the side lane of the unit AFTER this one may still be reading as its weight-gradient input -- found by
      // tests/test_gpu_paths.py::test_gpu_overlap2_weight_gradient_lane_is_deterministic on the first graph replay)
      const int kf = (d.kind == CSN_UNIT_GOCT && !fuse_apply && !c.lanes) ? ub.adj_fused[j] : -1;
      float* adjp = kf >= 0 ? reinterpret_cast<float*>(scratch + ub.adj[kf].off) : nullptr;
      const int st = run_bn_bwd(b, ui, u, j, fuse_apply, &bnargs[j], adjp, 1);
      if (st != CSN_OK) return st;
      bn_fused[j] = fuse_apply;
      bd.dz[j] = kf >= 0 ? c.act_y(d.out_act[j]) : reinterpret_cast<const float*>(c.ws + P.tz_off[d.out_act[j]]);
      if (kf >= 0) bd.adj[kf] = adjp;
    }
    {   // one finalise launch for the unit's branches, then the apply passes
      BnBwdArgs fin[CSN_MAX_BRANCH];
      int nfin = 0;
      for (int j = 0; j < d.n_out; ++j)
        if (d.cout[j] > 0) fin[nfin++] = bnargs[j];
      LAUNCH_TRY(csn_launch_bn_bwd_finalize_n(fin, nfin, c.stream));
      for (int j = 0; j < d.n_out; ++j)
        if (d.cout[j] > 0) LAUNCH_TRY(csn_launch_bn_bwd_apply_step(bnargs[j], c.stream));
    }
    bool c3q_fwd = false;   // the train-mode forward ran this 3x3 unit on c3q_kernel: its max-pooled input copies exist (run_unit)
    if (d.kind == CSN_UNIT_GOCT && d.ksize == 3 && P.c3q && P.tiled3)
      for (const PwLaunchPlan& L : u.pwl) c3q_fwd = c3q_fwd || L.c3q;
    for (int i = 0; i < d.n_in; ++i) {
      if (d.cin[i] == 0) continue;
      bd.in[i] = (d.kind == CSN_UNIT_GOCT && d.stride == 2 && !u.std_conv) ? reinterpret_cast<const float*>(c.ws + u.pooled_off[i])
                                                             : c.act_in(d.in_act[i]);
      if (c3q_fwd && u.mp_off[i] >= 0) bd.mp[i] = reinterpret_cast<const float*>(c.ws + u.mp_off[i]);
    }
  }
  for (int i = 0; i < d.n_in; ++i) {
    if (!ub.need_dx[i]) continue;
    float* g = grad_buf(c, d.in_act[i], u.in_slot[i]);
    bd.dx[i] = ub.dxp_off[i] >= 0 ? reinterpret_cast<float*>(scratch + ub.dxp_off[i]) : g;
    bd.dxsrc[i] = bd.dx[i];
  }
  if (ub.tmp_off >= 0) bd.tmp = reinterpret_cast<float*>(scratch + ub.tmp_off);

  // Weight gradients leave the critical path: they only read dz (final once the BN backward above has run) and saved
  // activations and nothing reads them before the optimizer, so they go to a side lane (own partial buffers) while the
  // caller's stream carries on with the input gradients and the next unit's BN backward.  Joined at the end of csn_backward.
  Ctx cs = c;
  if (c.lanes) {
    const int stf = lanes_fork(c, 1);
    if (stf != CSN_OK) return stf;
    cs.stream = P.lane[0];
    cs.side = true;
  }
  const BwdCtx bs{cs, b.arena, b.grad, b.flop_w, b.pen_scale, c.lanes ? nullptr : b.defer};
  if (d.kind == CSN_UNIT_DW) {
    DwArgs a;
    a.nbr = 0; a.B = S; a.a16 = c.a16 ? 1 : 0; a.variant = 0;
    int blk = 0;
    for (int k = 0; k < d.n_in; ++k) {
      if (d.cout[k] == 0) continue;
      const Act& act = P.acts[d.in_act[k]];
      const int H = P.H >> act.lvl, W = P.W >> act.lvl;
      DwWgradArgs w;
      w.a16 = c.a16 ? 1 : 0;
      w.dz = bd.dz[k]; w.x = bd.in[k]; w.partial = reinterpret_cast<double*>(c.ws + (cs.side ? P.red2_off : P.red_off)); w.grad = b.grad;
      w.off_w = d.w_off[k]; w.C = d.cout[k]; w.S = S; w.H = H; w.W = W; w.nslab = 0;
      // one-pass kernel (see dw_fused_slabs); the branches go one after the other (they share the partial buffer)
      const bool virt = dw_in_virtual(P, ui, k);
      const int fslabs = dw_fused_slabs(P, ub, k, act.lvl, cs.side, virt);
      if (fslabs > 0) {
        const Ctx& cf = (virt && cs.side) ? c : cs;   // (a never-stored input pins the unit to this kernel, on the caller's stream)
        if (&cf == &c) w.partial = reinterpret_cast<double*>(c.ws + P.red_off);
        const bool defer_fin = b.defer != nullptr && !cf.side && u.dwwg_off[k] >= 0;
        if (defer_fin) w.partial = reinterpret_cast<double*>(c.ws + u.dwwg_off[k]);   // own region: finalised with all the others
        DwArgs f;
        f.nbr = 1; f.B = S; f.a16 = c.a16 ? 1 : 0; f.variant = P.dwb_fast ? (P.dw_xl ? 2 : 1) : 0;
        DwBranch& fb = f.br[0];
        fb.in = bd.dz[k]; fb.out = bd.dx[k]; fb.xin = bd.in[k];
        if (virt) {   // x = PReLU(BN(z of the producer)) on load
          const int ia = d.in_act[k];
          const UnitPlan& pu = P.units[P.act_prod_unit[ia]];
          const int pj = P.act_prod_branch[ia];
          fb.xin = reinterpret_cast<const float*>(c.ws + P.tz_off[ia]);
          fb.in_scale = c.pk(pu.out_epi[pj].scale); fb.in_shift = c.pk(pu.out_epi[pj].shift); fb.in_alpha = c.pk(pu.out_epi[pj].alpha);
          fb.in_mean = c.pk(pu.tr_mean[pj]); fb.in_invstd = c.pk(pu.tr_invstd[pj]);
          fb.bnred = reinterpret_cast<double*>(c.ws + u.bnred_off[k]);
        }
        if (bn_fused[k]) {   // dz from dy and z on load
          const BnBwdArgs& ba = bnargs[k];
          fb.in = ba.dyA; fb.dy2 = ba.dyB; fb.zraw = ba.z;
          fb.bn_scale = ba.scale; fb.bn_shift = ba.shift; fb.bn_alpha = ba.alpha; fb.bn_mean = ba.mean; fb.bn_invstd = ba.invstd;
          fb.bn_m1m2 = ba.m1m2; fb.bn_gamma = ba.arena + ba.off_weight;
        }
        fb.w9 = c.pk(ub.dwf_w[k]);
        fb.scale = c.pk(P.ident.scale); fb.shift = c.pk(P.ident.shift); fb.alpha = c.pk(P.ident.alpha);
        fb.w9b = fb.scale_b = fb.shift_b = fb.alpha_b = nullptr;
        fb.pool = nullptr; fb.skip_out = 0; fb.stats = w.partial;
        fb.C = d.cout[k]; fb.H = H; fb.W = W;
        const int fcols = (W + 3) / 4;
        fb.LX = dw_lanes_x(fcols, P.dw_xl);
        fb.NY = CSN_BLOCK / fb.LX;
        fb.tiles_x = (fcols + fb.LX - 1) / fb.LX;
        fb.R = choose_dw_rows(H, fb.NY);
        fb.tiles_y = (H + fb.NY * fb.R - 1) / (fb.NY * fb.R);
        fb.blk_end = fb.tiles_x * fb.tiles_y * fb.C * S;
        LAUNCH_TRY(csn_launch_dw_bwd(f, cf.stream));
        if (bn_fused[k] && P.debug_dz) LAUNCH_TRY(csn_launch_bn_bwd_apply(bnargs[k], cf.stream));   // probes only: dz over z, afterwards
        w.nslab = fslabs;                                   // partials are there: finalise only
        if (defer_fin) {
          DwFinJob fj;
          fj.partial = w.partial; fj.off_w = w.off_w; fj.C = w.C; fj.nslab = fslabs;
          b.defer->dwfin.push_back(fj);
        } else {
          LAUNCH_TRY(csn_launch_dw_wgrad(w, cf.stream));
        }
        continue;
      }
      LAUNCH_TRY(csn_launch_dw_wgrad(w, cs.stream));
      if (!ub.need_dx[k]) continue;
      DwBranch& br = a.br[a.nbr++];
      br.in = bd.dz[k]; br.out = bd.dx[k];
      br.w9 = c.pk(ub.dwf_w[k]);
      br.scale = c.pk(P.ident.scale); br.shift = c.pk(P.ident.shift); br.alpha = c.pk(P.ident.alpha);
      br.w9b = br.scale_b = br.shift_b = br.alpha_b = nullptr;
      br.pool = nullptr; br.skip_out = 0; br.stats = nullptr; br.xin = nullptr;
      br.C = d.cout[k]; br.H = H; br.W = W;
      const int cols = (br.W + 3) / 4;
      br.LX = dw_lanes_x(cols, P.dw_xl);
      br.NY = CSN_BLOCK / br.LX;
      br.tiles_x = (cols + br.LX - 1) / br.LX;
      br.R = choose_dw_rows(br.H, br.NY);
      br.tiles_y = (br.H + br.NY * br.R - 1) / (br.NY * br.R);
      blk += br.tiles_x * br.tiles_y * br.C * S;
      br.blk_end = blk;
    }
    if (a.nbr > 0) LAUNCH_TRY(csn_launch_dw(a, c.stream));
    return CSN_OK;
  }

  // adjoint-upsampled dz (gOctConv low->high terms)
  for (size_t k = 0; k < ub.adj.size(); ++k) {
    const AdjPlan& ap = ub.adj[k];
    if (bd.adj[k] != nullptr) continue;   // written by the BatchNorm backward above (UnitBwd::adj_fused, not in side-lane mode)
    AdjUpArgs ua;
    ua.in = bd.dz[ap.j]; ua.out = reinterpret_cast<float*>(scratch + ap.off);
    ua.planes = S * ap.C; ua.Hl = P.H >> ap.lvl; ua.Wl = P.W >> ap.lvl; ua.f = ap.f;
    ua.in16 = ua.out16 = c.a16 ? 1 : 0;
    LAUNCH_TRY(csn_launch_adjup(ua, c.stream));
    bd.adj[k] = ua.out;
  }
  for (const WgPlan& w : ub.wg) {
    // a pass that reads the adjoint-upsampled dz lives in the shared scratch, which the next unit overwrites: main lane
    bool scratch_free = true;
    for (const WgRowSrc& r : w.rows) scratch_free = scratch_free && r.kind != SRC_ADJ;
    for (int s2 = 0; s2 < w.L.passes[0].nsrc; ++s2)
      scratch_free = scratch_free && (w.L.passes[0].src_kind[s2] == SRC_IN || w.L.passes[0].src_kind[s2] == SRC_DZ);
    const int st = run_wgrad(scratch_free ? bs : b, w, bd);
    if (st != CSN_OK) return st;
  }
  for (int i = 0; i < d.n_in; ++i)
    if (ub.need_dx[i] && ub.zero_dx[i]) {
      const Act& A = P.acts[d.in_act[i]];
      const int lvl = ub.dxp_off[i] >= 0 ? u.base_lvl + i : A.lvl;
      HIP_TRY(hipMemsetAsync(bd.dx[i], 0, (size_t)S * d.cin[i] * (P.H >> lvl) * (P.W >> lvl) * (c.a16 ? 2 : 4),
                             (hipStream_t)c.stream));
    }
  if (ub.msdx_ng > 0) {
    MsDxArgs ma;
    ma.dz = bd.dz[0]; ma.dx = bd.dx[0];
    int base = 0;
    for (int k = 0; k < CSN_NDIL; ++k) {
      ma.dch[k] = d.dil_ch[k]; ma.cobase[k] = base; base += d.dil_ch[k];
      ma.w[k] = d.dil_ch[k] ? c.pk(ub.msdx_w[k]) : nullptr;
    }
    ma.cin = d.cin[0]; ma.cout = d.cout[0]; ma.H = P.H >> u.base_lvl; ma.W = P.W >> u.base_lvl; ma.B = S;
    ma.ng = ub.msdx_ng; ma.a16 = c.a16 ? 1 : 0; ma.pad = 0;
    LAUNCH_TRY(csn_launch_ms_dx(ma, c.stream));
  }
  // DataLaunch::route: the direct chunks of a branch route the pooled term themselves where ALL of them run on pwq_kernel over rows
  // of a multiple of four elements (decided per branch: a mixed set would leave rows unrouted or route them twice)
  bool fuse_br[CSN_MAX_BRANCH] = {false, false, false};
  for (int i = 0; i < d.n_in; ++i) {
    int n = 0;
    bool ok = bd.in[i] != nullptr && bd.tmp != nullptr;
    for (const DataLaunch& dl : ub.data)
      if (dl.i == i && dl.route && !dl.pooled) {
        ++n;
        const int Wp = P.W >> dl.L.lvl, Hp = P.H >> dl.L.lvl;
        const bool on_pwq = P.pw4 && dl.L.pwq && (Wp & 3) == 0 && (Hp & 1) == 0;                       // (launch_pw's own predicates)
        const bool on_c3q = !dl.L.pwq && P.c3q && P.tiled3 && dl.L.c3q && !dl.L.c3q_z && (!c.a16 || c.raw) && (Wp & 1) == 0 && (Hp & 1) == 0;
        ok = ok && dl.L.passes.size() == 1 && (on_pwq || on_c3q);
      }
    fuse_br[i] = ok && n > 0;
  }
  for (const DataLaunch& dl : ub.data) {
    const bool fuse = dl.route && !dl.pooled && fuse_br[dl.i];
    PwBind bq = bd;
    if (fuse) { bq.route_x = bd.in[dl.i]; bq.route_t = bd.tmp; }
    const int st = launch_pw(c, dl.L, bq);
    if (st != CSN_OK) return st;
    if ((dl.to_tmp && !dl.route) || (dl.route_last && !fuse)) {
      PoolBwdArgs pa;
      pa.x = bd.in[dl.i]; pa.t = bd.tmp; pa.dx = bd.dx[dl.i];
      pa.planes = S * d.cin[dl.i]; pa.Hl = P.H >> dl.lvl_lo; pa.Wl = P.W >> dl.lvl_lo; pa.f = dl.pool_f; pa.a16 = c.a16 ? 1 : 0;
      LAUNCH_TRY(csn_launch_maxpool_bwd_add(pa, c.stream));
    }
  }
  if (d.kind == CSN_UNIT_GOCT && d.stride == 2 && !u.std_conv)
    for (int i = 0; i < d.n_in; ++i) {
      if (!ub.need_dx[i]) continue;
      PoolBwdArgs pa;
      pa.x = nullptr; pa.t = bd.dx[i]; pa.dx = grad_buf(c, d.in_act[i], u.in_slot[i]);
      pa.planes = S * d.cin[i]; pa.Hl = P.H >> (u.base_lvl + i); pa.Wl = P.W >> (u.base_lvl + i); pa.f = 2; pa.a16 = c.a16 ? 1 : 0;
      LAUNCH_TRY(csn_launch_avgpool2_bwd(pa, c.stream));
    }
  return CSN_OK;
}

}  // namespace

csn_plan::~csn_plan() = default;

extern "C" {

static int enable_training_impl(csn_plan* P);

// Transactional: everything the planning below lays out -- the bf16 re-layout of the workspace regions included -- is
// committed only when the whole call succeeds; a failure (an unsupported graph, a backward plan that does not fit) leaves the
// plan exactly as csn_plan_create built it, so that eval forwards and a later attempt see consistent offsets.
int csn_plan_enable_training(csn_plan* P) {
  if (!P) return CSN_E_INVALID;
  if (P->train) return CSN_OK;
  if (P->S != P->B) { g_hip_err = "training needs the whole batch in one slice (sub_batch = 0)"; return CSN_E_UNSUPPORTED; }
  struct Snapshot {
    std::vector<Act> acts; std::vector<UnitPlan> units; std::vector<csn_plan::WsAlloc> ws_allocs; std::vector<CsnPrepJob> jobs;
    int64_t ws_bytes, packed_floats, pen_off; bool act_half;
  } snap{P->acts, P->units, P->ws_allocs, P->jobs, P->ws_bytes, P->packed_floats, P->pen_off, P->act_half};
  const int st = enable_training_impl(P);
  if (st != CSN_OK && P->packed != nullptr) {   // (packed == nullptr: the device buffers themselves are gone -- nothing to keep consistent)
    P->acts = std::move(snap.acts); P->units = std::move(snap.units); P->ws_allocs = std::move(snap.ws_allocs);
    P->jobs = std::move(snap.jobs);
    P->ws_bytes = snap.ws_bytes; P->packed_floats = snap.packed_floats; P->pen_off = snap.pen_off; P->act_half = snap.act_half;
    P->tz_off.clear(); P->tg_off.clear(); P->n_cons.clear(); P->virt_cons.clear(); P->act_prod_unit.clear();
    P->act_prod_branch.clear(); P->orphan_acts.clear(); P->bwd.clear();
    P->train = false;
  }
  return st;
}

static int enable_training_impl(csn_plan* P) {
  Builder bl(*P);
  const int na = (int)P->acts.size(), nu = (int)P->units.size();
  if (P->act16) {
    // CSN_OPT_TRAIN_BF16 is already set: re-lay the regions csn_plan_create allocated with 2-byte activation elements (the
    // same allocation order; every field that holds one of the old offsets is mapped) and size everything below accordingly
    std::vector<csn_plan::WsAlloc> old = P->ws_allocs;
    P->ws_allocs.clear();
    P->ws_bytes = 0;
    P->act_half = true;
    std::vector<std::pair<int64_t, int64_t>> map;
    for (const csn_plan::WsAlloc& r : old) map.push_back({r.off, bl.alloc_ws(r.act ? r.bytes / 2 : r.bytes, r.act)});
    auto mv = [&](int64_t& off) {
      if (off < 0) return;
      for (const auto& m : map)
        if (m.first == off) { off = m.second; return; }
      off = -2;   // (not a region of the workspace: caught below)
    };
    bool ok = true;
    for (int i = 1; i < na; ++i) { mv(P->acts[i].ws_off); ok = ok && P->acts[i].ws_off != -2; }
    mv(P->pen_off); ok = ok && P->pen_off != -2;
    for (UnitPlan& u : P->units) {
      for (int i = 0; i < 3; ++i) {
        mv(u.pooled_off[i]); mv(u.mp_off[i]); mv(u.stats_off[i]); mv(u.gap_off[i]);
        ok = ok && u.pooled_off[i] != -2 && u.mp_off[i] != -2 && u.stats_off[i] != -2 && u.gap_off[i] != -2;
      }
      mv(u.z_off); mv(u.logits_off);
      ok = ok && u.z_off != -2 && u.logits_off != -2;
    }
    if (!ok) { g_hip_err = "workspace re-layout for bfloat16 tensors: unknown region"; return CSN_E_INVALID; }
  }
  P->tz_off.assign(na, -1);
  P->tg_off.assign(na, std::array<int64_t, 2>{-1, -1});
  P->n_cons.assign(na, 0);
  int maxc = 1;
  for (int i = 1; i < na; ++i) {
    P->tz_off[i] = bl.alloc_act(P->acts[i].channels, P->acts[i].lvl);
    maxc = std::max(maxc, P->acts[i].channels);
  }
  for (int k = 0; k < nu; ++k) {
    UnitPlan& u = P->units[k];
    for (int i = 0; i < u.d.n_in; ++i) {
      const int a = u.d.in_act[i];
      if (u.d.cin[i] == 0 || !wants_dx(*P, a)) continue;
      if (P->n_cons[a] >= 2) { g_hip_err = "an activation with more than two consumers"; return CSN_E_UNSUPPORTED; }
      u.in_slot[i] = P->n_cons[a]++;
      P->tg_off[a][u.in_slot[i]] = bl.alloc_act(P->acts[a].channels, P->acts[a].lvl);
    }
    if (u.d.kind == CSN_UNIT_CLS) continue;
    for (int j = 0; j < u.d.n_out; ++j) {
      if (u.d.cout[j] == 0) continue;
      u.tr_mean[j] = bl.alloc_packed(u.d.cout[j]);
      u.tr_invstd[j] = bl.alloc_packed(u.d.cout[j]);
      u.tr_m1m2[j] = bl.alloc_packed(2 * u.d.cout[j]);
    }
  }
  for (int k = 0; k < nu; ++k) {
    const UnitPlan& u = P->units[k];
    if (u.d.kind == CSN_UNIT_CLS) continue;
    for (int j = 0; j < u.d.n_out; ++j)
      if (u.d.cout[j] > 0 && P->n_cons[u.d.out_act[j]] == 0) {
        // an output nobody reads (e.g. a CSFHead.fuse branch whose MSBlock was pruned away): its gradient is zero, as
        // autograd would have it -- one zero-filled gradient buffer, cleared at the start of every backward
        const int a = u.d.out_act[j];
        P->n_cons[a] = 1;
        P->tg_off[a][0] = bl.alloc_act(P->acts[a].channels, P->acts[a].lvl);
        P->orphan_acts.push_back(a);
      }
  }
  P->x16_off = bl.alloc_ws((int64_t)P->S * P->acts[0].channels * P->H * P->W * 2);   // bf16 copy of the input batch (CSN_OPT_TRAIN_BF16)
  P->red_maxc = maxc;
  P->red_off = bl.alloc_ws((int64_t)maxc * CSN_BN_NSLAB * 9 * sizeof(double));
  P->red2_off = bl.alloc_ws((int64_t)maxc * CSN_BN_NSLAB * 9 * sizeof(double));   // ... of the weight-gradient side lane
  P->bwd.clear();
  P->bwd.resize(nu);
  int64_t scratch = 0, wg_floats = 0;
  for (int k = 0; k < nu; ++k) {
    UnitPlan& u = P->units[k];
    UnitBwd& ub = P->bwd[k];
    int st = CSN_OK;
    switch (u.d.kind) {
      case CSN_UNIT_GOCT: st = plan_goct_bwd(bl, u, ub); break;
      case CSN_UNIT_DW: st = plan_dw_bwd(bl, u, ub); break;
      case CSN_UNIT_MS: st = plan_ms_bwd(bl, u, ub); break;
      case CSN_UNIT_CLS: st = plan_cls_bwd(bl, u, ub); break;
      default: st = CSN_E_INVALID;
    }
    if (st != CSN_OK) {
      g_hip_err = "backward plan of unit " + std::to_string(k) + ": " + g_why;
      return st;
    }
    scratch = std::max(scratch, ub.scratch);
    for (const WgPlan& w : ub.wg) {
      const PwPassPlan& pp = w.L.passes[0];
      const int rows = std::min(pp.nrows, 128);   // (a launch's row chunk: <= 80 on the generic kernels, <= 128 on wgrad_bf16_kernel)
      wg_floats = std::max(wg_floats, (int64_t)WG_MAX_BLOCKS * ((rows + 15) & ~15) * ((pp.K + 15) & ~15));
    }
  }
  // activations that are never stored (csn_plan::virt_cons): only consumer = a depthwise unit whose forward leaves statistics
  // partials per (image, tile) and whose backward is the one-pass kernel with the BatchNorm apply fused in
  P->virt_cons.assign(na, -1);
  P->act_prod_unit.assign(na, -1);
  P->act_prod_branch.assign(na, -1);
  for (int k = 0; k < nu; ++k) {
    const UnitPlan& u = P->units[k];
    if (u.d.kind == CSN_UNIT_CLS) continue;
    for (int j = 0; j < u.d.n_out; ++j)
      if (u.d.cout[j] > 0) { P->act_prod_unit[u.d.out_act[j]] = k; P->act_prod_branch[u.d.out_act[j]] = j; }
  }
  if (P->bn_fwd_fuse && P->bn_bwd_fuse && !std::getenv("CSN_DW_BWD_SPLIT"))
    for (int k = 0; k < nu; ++k) {
      UnitPlan& u = P->units[k];
      if (u.d.kind != CSN_UNIT_DW) continue;
      bool ok = true;
      for (int i = 0; i < u.d.n_in && ok; ++i) {
        if (u.d.cout[i] == 0) continue;
        const int a = u.d.in_act[i];
        ok = a > 0 && P->n_cons[a] == 1 && P->act_prod_unit[a] >= 0 && P->bwd[k].need_dx[i] &&
             dw_stats_slabs(*P, P->acts[a].lvl) > 0;
        if (ok) {
          const int pk = P->units[P->act_prod_unit[a]].d.kind;
          ok = pk == CSN_UNIT_GOCT || pk == CSN_UNIT_DW;
        }
      }
      if (!ok) continue;
      for (int i = 0; i < u.d.n_in; ++i) {
        if (u.d.cout[i] == 0) continue;
        P->virt_cons[u.d.in_act[i]] = k;
        u.gapin_off[i] = bl.alloc_ws((int64_t)u.d.cout[i] * CSN_BN_NSLAB * sizeof(double));
        u.bnred_off[i] = bl.alloc_ws((int64_t)u.d.cout[i] * CSN_BN_NSLAB * 3 * sizeof(double));
      }
    }
  // bf16 mode: the 1x1 units on pw4_kernel leave their outputs' statistics partials themselves (Pw4Args::stats_h), one slab per
  // (image, item tile) of the launch that stores the branch
  if (P->act16 && P->pw4 && P->pw4_stats)
    for (int k = 0; k < nu; ++k) {
      UnitPlan& u = P->units[k];
      if (u.d.kind != CSN_UNIT_GOCT || !u.pw4 || u.pw4l.empty()) continue;
      int writers[CSN_MAX_BRANCH] = {0, 0, 0};
      for (const UnitPlan::Pw4Launch& L : u.pw4l)
        for (int j : {L.hi_out, L.lo_out})
          if (j >= 0) ++writers[j];
      for (const UnitPlan::Pw4Launch& L : u.pw4l) {
        int twl, tx, ty;
        pw4_tile_geo(*P, P->H >> (u.base_lvl + L.bl), P->W >> (u.base_lvl + L.bl), &twl, &tx, &ty);
        const int64_t n = (int64_t)P->S * tx * ty;
        if (n > (1 << 20) || !csn_pw4_has_stats(L.nth, L.ntl)) continue;
        for (int j : {L.hi_out, L.lo_out}) {
          if (j < 0 || u.d.cout[j] == 0 || ((u.pw4_old_mask >> j) & 1) || writers[j] != 1) continue;
          u.pstats_off[j] = bl.alloc_ws((int64_t)u.d.cout[j] * n * 2 * sizeof(double));
          u.pstats_n[j] = (int)n;
        }
      }
    }
  P->scratch_bytes = scratch;
  P->scratch_off = bl.alloc_ws(scratch > 0 ? scratch : 256);
  P->wg_region_floats = wg_floats;
  P->wg_off = bl.alloc_ws(wg_floats * (int64_t)sizeof(float) * CSN_WG_REGIONS);
  P->wg2_off = bl.alloc_ws(wg_floats * (int64_t)sizeof(float));
  for (int k = 0; k < nu; ++k) {
    UnitPlan& u = P->units[k];
    if (u.d.kind != CSN_UNIT_DW) continue;
    for (int i = 0; i < u.d.n_in; ++i)
      if (u.d.cout[i] > 0) u.dwwg_off[i] = bl.alloc_ws((int64_t)u.d.cout[i] * CSN_BN_NSLAB * 9 * sizeof(double));
  }
  // the packed buffer and the job list grew: re-allocate / re-upload
  if (P->packed) (void)hipFree(P->packed);
  if (P->jobs_dev) (void)hipFree(P->jobs_dev);
  P->packed = nullptr; P->jobs_dev = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&P->packed), (size_t)(P->packed_floats + 4) * sizeof(float));
  if (e != hipSuccess) { hip_fail(e, "hipMalloc(packed)"); return CSN_E_NOMEM; }
  e = hipMemsetAsync(P->packed, 0, (size_t)(P->packed_floats + 4) * sizeof(float), nullptr);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&P->jobs_dev), P->jobs.size() * sizeof(CsnPrepJob));
  if (e == hipSuccess)
    e = hipMemcpy(P->jobs_dev, P->jobs.data(), P->jobs.size() * sizeof(CsnPrepJob), hipMemcpyHostToDevice);
  if (e != hipSuccess) return hip_fail(e, "plan upload");
  P->params_ready = false;
  P->train = true;
  drop_graph(P);
  return CSN_OK;
}

int csn_plan_train_act_info(const csn_plan* P, int32_t id, csn_train_act_info* out) {
  if (!P || !out || id < 0 || id >= (int)P->acts.size()) return CSN_E_INVALID;
  if (!P->train) return CSN_E_STATE;
  out->act_offset_bytes = P->acts[id].ws_off;
  out->z_offset_bytes = P->tz_off[id];
  out->dz_offset_bytes = P->tz_off[id];
  {
    const int pu = id < (int)P->act_prod_unit.size() ? P->act_prod_unit[id] : -1;
    if (pu >= 0 && P->units[pu].d.kind == CSN_UNIT_GOCT && P->bwd[pu].adj_fused[P->act_prod_branch[id]] >= 0 &&
        !P->last_bwd_lanes)   // the decision the last csn_backward actually took (side lane: dz stays over z) -- ADVICE r4: asking the
                              // options again disagreed with it whenever the lanes could not be created (always so on the emulator)
      out->dz_offset_bytes = P->acts[id].ws_off;
  }
  out->grad_offset_bytes[0] = P->tg_off[id][0];
  out->grad_offset_bytes[1] = P->tg_off[id][1];
  out->x16_offset_bytes = (id == 0 && P->act16) ? P->x16_off : -1;
  out->n_consumers = P->n_cons[id];
  out->bf16 = P->act16 ? 1 : 0;
  return CSN_OK;
}

int32_t csn_plan_unit_in_slot(const csn_plan* P, int32_t unit, int32_t branch) {
  if (!P || !P->train || unit < 0 || unit >= (int)P->units.size() || branch < 0 || branch >= CSN_MAX_BRANCH) return -1;
  const UnitPlan& u = P->units[unit];
  if (branch >= u.d.n_in || u.d.cin[branch] == 0 || !wants_dx(*P, u.d.in_act[branch])) return -1;
  return u.in_slot[branch];
}

int csn_backward(csn_plan* P, const float* x, const float* dy, void* workspace, const float* arena, float* grad,
                 int64_t arena_floats, const float* flop_w, float pen_scale, void* stream) {
  if (!P || !x || !dy || !workspace || !arena || !grad || !flop_w) return CSN_E_INVALID;
  if (!P->train || !P->params_ready || !P->bn_tables_train) return CSN_E_STATE;   // needs csn_forward_train first
  (void)arena_floats;
  uint32_t ps_bits;
  std::memcpy(&ps_bits, &pen_scale, 4);
  const std::vector<uint64_t> key = {(uint64_t)(uintptr_t)x, (uint64_t)(uintptr_t)dy, (uint64_t)(uintptr_t)workspace,
                                     (uint64_t)(uintptr_t)arena, (uint64_t)(uintptr_t)grad, (uint64_t)ps_bits,
                                     hash_floats(flop_w, (int)P->units.size() * CSN_MAX_BRANCH)};
  return run_graphed(P, P->g_bwd, key, stream, [&](void* s) {
    Ctx c{*P, x, nullptr, static_cast<char*>(workspace), s};
    c.raw = true;
    c.a16 = P->act16;
    c.lanes = P->overlap_bwd && lanes_ready(P);
    P->last_bwd_lanes = c.lanes;   // measured: no gain for the train step (106.3 vs 105.5 ms), off by default
    BwdDefer defer;
    const BwdCtx b{c, arena, grad, flop_w, pen_scale, (c.lanes || std::getenv("CSN_BWD_NO_DEFER")) ? nullptr : &defer};
    for (int a : P->orphan_acts)
      HIP_TRY(hipMemsetAsync(c.ws + P->tg_off[a][0], 0, (size_t)P->S * P->acts[a].channels * (P->H >> P->acts[a].lvl) *
                                                           (P->W >> P->acts[a].lvl) * (c.a16 ? 2 : 4), (hipStream_t)s));
    for (int u = (int)P->units.size() - 1; u >= 0; --u) {
      const int st = run_unit_bwd(b, u, dy);
      if (st != CSN_OK) return st;
    }
    if (c.lanes) {       // the weight-gradient lane rejoins the caller's stream
      const int st = lanes_join(c, 1);
      if (st != CSN_OK) return st;
    }
    if (b.defer) {
      const int st = flush_wgred(b);
      if (st != CSN_OK) return st;
      if (!defer.dwfin.empty()) {
        const int fe = csn_launch_dw_wgrad_finalize_batch(defer.dwfin.data(), (int)defer.dwfin.size(), grad, s);
        if (fe != 0) return (int)CSN_E_HIP;
      }
    }
    return (int)CSN_OK;
  });
}

int csn_bce_with_logits(const float* y, const float* t, float* dy, int64_t n, double* loss, void* stream) {
  if (!y || !t || !dy || !loss || n <= 0) return CSN_E_INVALID;
  LAUNCH_TRY(csn_launch_bce(y, t, dy, n, loss, stream));
  return CSN_OK;
}

int csn_saliency_u8(const float* logits, uint8_t* out, int64_t n, void* stream) {
  if (!logits || !out || n <= 0) return CSN_E_INVALID;
  LAUNCH_TRY(csn_launch_saliency_u8(logits, out, n, stream));
  return CSN_OK;
}

int csn_resize_normalize_nchw(const float* hwc, float* nchw, int32_t B, int32_t Hi, int32_t Wi, int32_t H, int32_t W, void* stream) {
  if (!hwc || !nchw || B <= 0 || Hi <= 0 || Wi <= 0 || H <= 0 || W <= 0) return CSN_E_INVALID;
  LAUNCH_TRY(csn_launch_resize_normalize(hwc, nchw, B, Hi, Wi, H, W, stream));
  return CSN_OK;
}

int csn_saliency_resize_u8(const float* logits, uint8_t* out, int32_t H, int32_t W, int32_t h, int32_t w, void* stream) {
  if (!logits || !out || H <= 0 || W <= 0 || h <= 0 || w <= 0) return CSN_E_INVALID;
  LAUNCH_TRY(csn_launch_saliency_resize_u8(logits, out, H, W, h, w, stream));
  return CSN_OK;
}

int csn_resize_bilinear(const float* in, float* out, int32_t planes, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, void* stream) {
  if (!in || !out || planes <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return CSN_E_INVALID;
  LAUNCH_TRY(csn_launch_resize_bilinear(in, out, planes, Hi, Wi, Ho, Wo, stream));
  return CSN_OK;
}

int csn_val_mae(const float* logits, int32_t hi, int32_t wi, const float* target, int32_t h, int32_t w, double* mae,
                void* stream) {
  if (!logits || !target || !mae || hi <= 0 || wi <= 0 || h <= 0 || w <= 0) return CSN_E_INVALID;
  LAUNCH_TRY(csn_launch_val_mae(logits, hi, wi, target, h, w, mae, stream));
  return CSN_OK;
}

int csn_normalize_nchw(const float* hwc, float* nchw, int64_t B, int64_t H, int64_t W, void* stream) {
  if (!hwc || !nchw || B <= 0 || H <= 0 || W <= 0) return CSN_E_INVALID;
  LAUNCH_TRY(csn_launch_normalize_nchw(hwc, nchw, B, H * W, stream));
  return CSN_OK;
}

int csn_sal_hist(const uint8_t* sal, const uint8_t* gt, int64_t npix, int32_t n_images, uint64_t* hist, uint64_t* abs_sum,
                 void* stream) {
  if (!sal || !gt || !hist || !abs_sum || npix <= 0 || n_images <= 0) return CSN_E_INVALID;
  LAUNCH_TRY(csn_launch_sal_hist(sal, gt, npix, n_images, reinterpret_cast<unsigned long long*>(hist),
                                 reinterpret_cast<unsigned long long*>(abs_sum), stream));
  return CSN_OK;
}

int csn_stream_copy(const float* src, float* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0 || (n & 3) != 0) return CSN_E_INVALID;
  LAUNCH_TRY(csn_launch_stream_copy(src, dst, n, stream));
  return CSN_OK;
}

int csn_adam_step(float* p, const float* g, float* m, float* v, const float* wd, int64_t n, float lr, float beta1,
                  float beta2, float eps, int32_t step, void* stream) {
  if (!p || !g || !m || !v || !wd || n <= 0 || step <= 0) return CSN_E_INVALID;
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.wd = wd; a.n = n;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
  a.step_size = (float)((double)lr / bc1);
  a.sqrt_bc2 = (float)std::sqrt(bc2);
  LAUNCH_TRY(csn_launch_adam(a, stream));
  return CSN_OK;
}

}  // extern "C"
