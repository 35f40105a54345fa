// qcnn_planner.hip — launch planner of the conv table kernels: tile selection per family (which channels per wave, which tile),
// cost models, and the decision which family runs a launch.  Host code only (see qcnn_planner.h); the kernels it plans for are
// GetInPdMat + CalcFeatMap_ConvAprx (src/CaffeEva.cc:1261-1296, :760-868) in their forms of qcnn_kernels.hip / qcnn_sym8.hip /
// qcnn_half8.hip.
#include "qcnn_planner.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <queue>
#include <utility>
#include <vector>

// ------------------------------------------------------------------------------------------------------------------
// tile selection
// ------------------------------------------------------------------------------------------------------------------
// Tile selection: the 12 gather waves split the channels of one group (qk_conv_slots: the workgroup covers
// all of them whenever 12 x 32 allow it — every further channel chunk would rebuild the same LUT stages); each
// wave then owns as many positions as 64-72 accumulator registers leave room for.  The MFMA builder is
// instantiated for K in {16, 32, 64, 128}; any other K <= 128 runs the exact builder.
bool qk_conv_sym_shape(int Cin, int grp, int Ct, int M, int Cs, int K) {
  if (grp < 1 || Ct % grp || Cin % grp) return false;
  const int Cg = Cin / grp;
  // exactly 16 waves x 8 channels, K = 128, every sub-space complete with 4 or 8 dims (no operand masks in k_conv_sym)
  return Ct / grp == 128 && K == 128 && (Cs == 4 || Cs == 8) && Cg % Cs == 0 && M == Cg / Cs;
}


Qk8Config qk_conv_sym8_config(int Cin, int grp, int Ct, int M, int Cs, int K) {
  Qk8Config cf = {0, 0, 0, 0, 0};
  if (grp < 1 || Ct % grp || Cin % grp) return cf;
  const int Cg = Cin / grp, Ctg = Ct / grp;
  // K = 128, every sub-space complete with 4 or 8 dims (no operand masks in the kernel)
  if (K != 128 || !(Cs == 4 || Cs == 8) || Cg % Cs || M != Cg / Cs) return cf;
  // channels per wave x positions = 96: as many channels of the group in ONE workgroup as 8 waves hold (every further
  // channel chunk builds the same stages again), the tile that goes with it
  const int chunks = (Ctg + 383) / 384;
  const int per = (Ctg + chunks - 1) / chunks;
  if (per <= 64) return cf;                      // narrow layers: the sliding kernels of k_conv_aprx build less
  if (per <= 128) { cf.cpw = 16; cf.th = 2; cf.tw = 3; }
  else if (per <= 192) { cf.cpw = 24; cf.th = 2; cf.tw = 2; }
  else if (per <= 256) { cf.cpw = 32; cf.th = 1; cf.tw = 3; }
  else { cf.cpw = 48; cf.th = 1; cf.tw = 2; }
  if (Ctg % cf.cpw) { cf.cpw = 0; return cf; }   // a wave's channels all exist or none does
  cf.chunks = (Ctg + 8 * cf.cpw - 1) / (8 * cf.cpw);
  return cf;
}

size_t qk_conv_sym8_program_bytes(const Qk8Config& cf, int groups, int knl, int stride, int M) {
  if (!cf.cpw) return 0;
  const int rfH = cf.slide ? cf.th * stride : (cf.th - 1) * stride + knl, rfW = (cf.tw - 1) * stride + knl;
  return (size_t)rfH * rfW * M * groups * cf.chunks * 8 * cf.th * cf.tw * cf.cpw * sizeof(uint16_t);   // 16 half-waves x NP x CPW / 2
}


// Sliding form: th = slots = ceil(knl / stride) (3 or 5 built), tw = output columns of a strip; channels per wave as above
// for layers of up to 256 channels per workgroup (48 channels per wave would need 144 pairs for three slots).
Qk8Config qk_conv_sym8_slide_config(int Cin, int grp, int Ct, int M, int Cs, int K, int knl, int stride) {
  Qk8Config cf = {0, 0, 0, 0, 0};
  if (grp < 1 || Ct % grp || Cin % grp) return cf;
  const int Cg = Cin / grp, Ctg = Ct / grp;
  if (K != 128 || !(Cs == 4 || Cs == 8) || Cg % Cs || M != Cg / Cs) return cf;
  const int ns = (knl + stride - 1) / stride;
  const int chunks = (Ctg + 255) / 256;
  const int per = (Ctg + chunks - 1) / chunks;
  int cpw = 0, nc = 0;
  if (ns == 3) {
    if (per <= 64) return cf;
    if (per <= 128) { cpw = 16; nc = 2; } else if (per <= 192) { cpw = 24; nc = 1; } else { cpw = 32; nc = 1; }
  } else if (ns == 5) {
    if (per <= 64 || per > 128) return cf;
    cpw = 16; nc = 1;
  } else {
    return cf;
  }
  if (Ctg % cpw) return cf;
  cf.cpw = cpw; cf.th = ns; cf.tw = nc; cf.slide = 1;
  cf.chunks = (Ctg + 8 * cpw - 1) / (8 * cpw);
  return cf;
}


// Tiles of the fp16-sum form (QCNN_OPT_LUT_MODE = 3): the same channels per wave, twice the positions (192 pairs per wave)
Qk8Config qk_conv_sym8_config16(int Cin, int grp, int Ct, int M, int Cs, int K) {
  Qk8Config cf = qk_conv_sym8_config(Cin, grp, Ct, M, Cs, K);
  switch (cf.cpw) {
    case 48: cf.th = 2; cf.tw = 2; break;      // 384 channels: 2x2 (1x2 with fp32 sums)
    case 32: cf.th = 2; cf.tw = 3; break;      // 256: 2x3 (1x3)
    case 24: cf.th = 2; cf.tw = 4; break;      // 192: 2x4 (2x2)
    case 16: cf.th = 3; cf.tw = 4; break;      // 128: 3x4 (2x3)
    default: break;
  }
  return cf;
}


QkH8Config qk_conv_half8_config(int Cin, int grp, int Ct, int M, int Cs, int K) {
  QkH8Config cf = {0, 0, 0, 0, 0, 0};
  if (grp < 1 || Ct % grp || Cin % grp) return cf;
  const int Cg = Cin / grp, Ctg = Ct / grp;
  // K = 128, every sub-space complete with 4 or 8 dims (no operand masks in the kernel)
  if (K != 128 || !(Cs == 4 || Cs == 8) || Cg % Cs || M != Cg / Cs) return cf;
  // ONE workgroup holds all channels of a group (up to 512; more: chunks of equal size), the tile that fills 1536 sums
  const int chunks = (Ctg + 511) / 512;
  if (Ctg % chunks) return cf;
  switch (Ctg / chunks) {
    case 128: cf.cpw = 32; cf.th = 3; cf.tw = 4; cf.ws = 2; break;
    case 192: cf.cpw = 48; cf.th = 2; cf.tw = 4; cf.ws = 2; break;
    case 256: cf.cpw = 32; cf.th = 2; cf.tw = 3; cf.ws = 1; break;
    case 384: cf.cpw = 48; cf.th = 2; cf.tw = 2; cf.ws = 1; break;
    case 512: cf.cpw = 64; cf.th = 1; cf.tw = 3; cf.ws = 1; break;
    default: return cf;
  }
  cf.chunks = chunks;
  return cf;
}

size_t qk_conv_half8_program_bytes(const QkH8Config& cf, int groups, int knl, int stride, int M) {
  if (!cf.cpw) return 0;
  const int rfH = cf.slide ? cf.th * stride : (cf.th - 1) * stride + knl, rfW = (cf.tw - 1) * stride + knl;
  // per entry: groups x chunks x 8 waves x 4 lane groups x [positions per wave][cpw / 4] uint16
  return (size_t)rfH * rfW * M * groups * cf.chunks * 8 * 4 * (cf.th * cf.tw / cf.ws) * (cf.cpw / 4) * sizeof(uint16_t);
}


// Sliding form: th = slots = ceil(knl / stride) (3 built), tw = output columns of a strip: 128 channels per group 3 x 4 (two wave
// sets), 192: 3 x 2 (two sets, 144 of the 192 sums per wave), 256: 3 x 2, 384: 3 x 1 (144 sums), 512: 3 x 1
QkH8Config qk_conv_half8_slide_config(int Cin, int grp, int Ct, int M, int Cs, int K, int knl, int stride) {
  QkH8Config cf = qk_conv_half8_config(Cin, grp, Ct, M, Cs, K);
  if (!cf.cpw) return cf;
  const int ns = (knl + stride - 1) / stride;
  if (ns != 3) { cf.cpw = 0; return cf; }
  cf.th = 3; cf.slide = 1;
  switch ((Ct / grp) / cf.chunks) {
    case 128: cf.tw = 4; break;
    case 192: cf.tw = 2; break;
    case 256: cf.tw = 2; break;
    case 384: cf.tw = 1; break;
    case 512: cf.tw = 1; break;
    default: cf.cpw = 0; break;
  }
  return cf;
}


// ------------------------------------------------------------------------------------------------------------------
// cost models
// ------------------------------------------------------------------------------------------------------------------
// predicted duration (in stage-times, like QkSplitPlan::cost) of the symmetric kernel for a launch over p.panels panels:
// 2x2 tiles, list-scheduled heaviest first on 256 CUs
double qk_conv_sym_cost(const ConvParams& p) {
  const int tilesX = (p.Wo + 1) / 2, tilesY = (p.Ho + 1) / 2, tiles = tilesX * tilesY;
  std::vector<double> cu(256, 0.0);
  std::vector<double> cost((size_t)tiles);
  for (int r = 0; r < tiles; ++r) {
    int ty, tx;
    tile_of_rank(r, tilesY, tilesX, ty, tx);
    const int ho0 = ty * 2, wo0 = tx * 2;
    const int hoL = std::min(ho0 + 2, p.Ho) - 1, woL = std::min(wo0 + 2, p.Wo) - 1;
    const int rows = std::min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1) - std::max(0, ho0 * p.stride - p.pad) + 1;
    const int cols = std::min(p.W - 1, woL * p.stride - p.pad + p.knl - 1) - std::max(0, wo0 * p.stride - p.pad) + 1;
    // a symmetric stage serves a quarter more look-ups than the 1x3 tile's and takes longer; the factor is calibrated on
    // AlexNet conv2 so that the planner's choice matches the measurements (1000 / 500 / 250 images: symmetric -5.9 / -4.2 /
    // -2.0 %, 125 images: +17 %)
    cost[r] = 1.09 * ((double)std::max(rows, 0) * std::max(cols, 0) * p.M) + 10.0;
  }
  const long long wgs = (long long)tiles * p.panels * p.grp;
  if (wgs >= 8 * 256) {
    double sum = 0.0;
    for (int r = 0; r < tiles; ++r) sum += cost[r];
    return sum * p.panels * p.grp / 256.0;
  }
  // dispatch order: rank-major, panels and groups inside
  std::priority_queue<double, std::vector<double>, std::greater<double>> q;
  for (int i = 0; i < 256; ++i) q.push(0.0);
  double end = 0.0;
  for (int r = 0; r < tiles; ++r)
    for (int k = 0; k < p.panels * p.grp; ++k) {
      const double t = q.top() + cost[r];
      q.pop(); q.push(t);
      end = std::max(end, t);
    }
  return end;
}



// Split plan of a conv launch (ConvParams::splitZ).  A workgroup occupies a CU for its tile's whole stage sequence
// (0.1 - 0.4 ms), so a launch of a few hundred workgroups — one GPU's share of a batch sharded over 4 - 8 GPUs — leaves
// CUs idle for whole tile durations.  List-schedule the launch (dispatch order, 256 CUs, cost = stages + a fixed part)
// for a few candidate splits — none; the last `r` tiles, r = what exceeds whole rounds of 256 workgroups; all tiles —
// and slice counts, add the cost of writing and re-reading the partial sums, keep the cheapest.
QkSplitPlan qk_conv_plan(const ConvParams& p, size_t scratchFloats) {
  const int Ctg = p.Ct / p.grp;
  const QkSlots sl = qk_conv_slots(Ctg, p.grp);
  int TH, TW;
  qk_conv_tile(sl.cpw, &TH, &TW);
  const int tilesX = (p.Wo + TW - 1) / TW, tilesY = (p.Ho + TH - 1) / TH, tiles = tilesX * tilesY;
  const int ny = sl.chunks * p.grp;
  const int G = qcnn_stage_group(p.K), MG = (p.M + G - 1) / G;
  QkSplitPlan none = {tiles, 1, 0, 0.0};
  if ((long long)tiles * p.panels * ny >= 8 * 256) {                     // enough workgroups for the tail not to matter
    double stages = 0.0;
    const int G0 = qcnn_stage_group(p.K), MG0 = (p.M + G0 - 1) / G0;
    for (int r = 0; r < tiles; ++r) {
      int ty, tx;
      tile_of_rank(r, tilesY, tilesX, ty, tx);
      const int ho0 = ty * TH, wo0 = tx * TW;
      const int hoL = std::min(ho0 + TH, p.Ho) - 1, woL = std::min(wo0 + TW, p.Wo) - 1;
      const int rows = std::min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1) - std::max(0, ho0 * p.stride - p.pad) + 1;
      const int cols = std::min(p.W - 1, woL * p.stride - p.pad + p.knl - 1) - std::max(0, wo0 * p.stride - p.pad) + 1;
      stages += (double)std::max(rows, 0) * std::max(cols, 0) * MG0 + 10.0;
    }
    none.cost = stages * p.panels * ny / 256.0;
    return none;
  }
  std::vector<int> S(tiles);
  for (int r = 0; r < tiles; ++r) {
    int ty, tx;
    tile_of_rank(r, tilesY, tilesX, ty, tx);
    const int ho0 = ty * TH, wo0 = tx * TW;
    const int hoL = std::min(ho0 + TH, p.Ho) - 1, woL = std::min(wo0 + TW, p.Wo) - 1;
    const int rows = std::min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1) - std::max(0, ho0 * p.stride - p.pad) + 1;
    const int cols = std::min(p.W - 1, woL * p.stride - p.pad + p.knl - 1) - std::max(0, wo0 * p.stride - p.pad) + 1;
    S[r] = std::max(rows, 0) * std::max(cols, 0) * MG;
  }
  const double kFixed = 10.0;          // stage-times a workgroup spends outside its stage loop (roles, first stage, stores)
  const double kStageUs = 1.1;         // ~2700 cycles
  std::vector<double> cu(256);
  auto makespan = [&](int splitFrom, int Z) {
    std::fill(cu.begin(), cu.end(), 0.0);
    std::make_heap(cu.begin(), cu.end(), std::greater<double>());
    auto run = [&](double cost) {
      std::pop_heap(cu.begin(), cu.end(), std::greater<double>());
      cu.back() += cost;
      std::push_heap(cu.begin(), cu.end(), std::greater<double>());
    };
    for (int y = 0; y < ny; ++y) {               // dispatch order: x fastest
      for (int r = 0; r < splitFrom; ++r)
        for (int pn = 0; pn < p.panels; ++pn) run(S[r] + kFixed);
      for (int r = splitFrom; r < tiles; ++r)
        for (int z = 0; z < Z; ++z)
          for (int pn = 0; pn < p.panels; ++pn) run((double)S[r] / Z + kFixed);
    }
    return *std::max_element(cu.begin(), cu.end());
  };
  // stage-times of the reduction: k_conv_sum reads Z slabs and writes one at ~4 TB/s behind a launch; the Z slab stores of
  // the conv kernel itself mostly hide under other workgroups' stages (calibrated on conv3 / conv5 of AlexNet, one panel:
  // predicted 36 / 24 stage-times, measured 35 / 27)
  auto reduceCost = [&](int splitFrom, int Z) {
    const double slab = (double)(tiles - splitFrom) * p.panels * TH * TW * p.Ct * QCNN_PANEL * 4.0;
    return (slab * (Z + 1.0) / 4.0e6 + slab * Z / 10.0e6 + 5.0) / kStageUs;
  };
  QkSplitPlan best = none;
  double bestCost = makespan(tiles, 1);
  best.cost = bestCost;
  const long long wgs = (long long)tiles * p.panels * ny;
  const int rem = (int)(wgs % 256);                 // workgroups beyond whole rounds
  // candidate tails: every tile; the tiles beyond whole rounds of 256 workgroups; that tail widened by a quarter, a half
  // and a whole round (finer slices at the end of the launch balance the last round better)
  std::vector<int> cand = {0};
  if (rem > 0 && wgs > 256) {
    const int perTile = p.panels * ny;
    for (int extra : {0, 64, 128, 256}) {
      const int from = tiles - (rem + extra + perTile - 1) / perTile;
      if (from > 0 && std::find(cand.begin(), cand.end(), from) == cand.end()) cand.push_back(from);
    }
  }
  for (const int from : cand) {
    if (from >= tiles) continue;
    int minS = S[from];
    for (int r = from; r < tiles; ++r) minS = std::min(minS, S[r]);
    for (int Z = 2; Z <= 8; ++Z) {
      if (minS < 6 * Z) break;                      // a slice keeps at least six stages
      const size_t need = (size_t)(tiles - from) * Z * p.panels * TH * TW * p.Ct * QCNN_PANEL;
      if (need > scratchFloats) break;
      const double c = makespan(from, Z) + reduceCost(from, Z);
      if (c < bestCost * 0.97) { bestCost = c; best.splitFrom = from; best.Z = Z; best.partialFloats = need; best.cost = c; }
    }
  }
  return best;
}

// Segments of the sliding variant.  A segment of L output rows sweeps (L - 1) * stride + knl source rows (clipped), i.e.
// it re-builds knl - stride rows of its upper neighbour's strip: few, long segments build the least, but a launch of
// columns x segments x groups x panels workgroups must also fill 256 CUs evenly.  Candidates: 1 .. 4 equal segments and
// "one long + one short" cuts; list-scheduled (longest first) like qk_conv_plan; taken when it beats the tile kernel.
double qk_conv_plan_slide(ConvParams& p, double tileCost) {
  p.nSeg = 0;
  const int Ctg = p.Ct / p.grp;
  const QkSlide sc = qk_slide_config(Ctg, p.grp, p.knl, p.stride);
  const int ns = sc.ns;
  if (ns == 0 || p.K != 128 || p.Ho < 2 * ns) return 0.0;
  const int ny = sc.sl.chunks * p.grp;               // every channel chunk builds the strip's stages again
  const int G = qcnn_stage_group(p.K), MG = (p.M + G - 1) / G;
  const double kFixed = 12.0;
  const int nc = sc.nc;                                // output columns per strip
  const int colGroups = (p.Wo + nc - 1) / nc;
  auto segStages = [&](int cgi, int a, int b) {        // strip of output columns [cgi * nc, ..), output rows [a, b)
    const int wA = cgi * nc, wB = std::min(p.Wo, wA + nc) - 1;
    const int cols = std::min(p.W - 1, wB * p.stride - p.pad + p.knl - 1) - std::max(0, wA * p.stride - p.pad) + 1;
    const int rows = std::min(p.H - 1, (b - 1) * p.stride - p.pad + p.knl - 1) - std::max(0, a * p.stride - p.pad) + 1;
    return (double)std::max(rows, 0) * std::max(cols, 0) * MG;
  };
  // a sliding stage costs about what a tile stage costs (measured 0.68 vs 0.74 us per stage-time of this model on
  // AlexNet conv1), and every source row ends with the store + restart of a slot.  Measured: AlexNet conv1 (11 stages per column) -10 %, conv5 (72) -15 %,
  // VGG-16 conv1_2 (24) -12 %, its 128-channel layers (24 / 48) -25 %, but conv1_1 (3 stages per column: one sub-space,
  // three rows) +47 % — a column must hold enough stages to carry its restart.
  if (std::min(p.knl + (sc.nc - 1) * p.stride, p.W) * MG < 6 && tileCost < 1e29) return 0.0;      // (forced mode, tests: slides anyway)
  auto segCost = [&](int wo, int a, int b) {
    const int rows = std::min(p.H - 1, (b - 1) * p.stride - p.pad + p.knl - 1) - std::max(0, a * p.stride - p.pad) + 1;
    return segStages(wo, a, b) + 0.3 * std::max(rows, 0);
  };
  std::vector<double> cu(256);
  auto makespan = [&](const std::vector<int>& beg) {        // beg: nSeg + 1 boundaries, segments sorted longest first
    std::fill(cu.begin(), cu.end(), 0.0);
    std::make_heap(cu.begin(), cu.end(), std::greater<double>());
    const int nSeg = (int)beg.size() - 1;
    for (int y = 0; y < ny; ++y)
      for (int sgi = 0; sgi < nSeg; ++sgi)
        for (int wo = 0; wo < colGroups; ++wo)
          for (int pn = 0; pn < p.panels; ++pn) {
            std::pop_heap(cu.begin(), cu.end(), std::greater<double>());
            cu.back() += segCost(wo, beg[sgi], beg[sgi + 1]) + kFixed;
            std::push_heap(cu.begin(), cu.end(), std::greater<double>());
          }
    return *std::max_element(cu.begin(), cu.end());
  };
  std::vector<std::vector<int> > cands;
  for (int n = 1; n <= 4 && n * ns <= p.Ho; ++n) {            // n (nearly) equal segments
    std::vector<int> b(n + 1);
    for (int i = 0; i <= n; ++i) b[i] = (int)(((long long)p.Ho * i + n - 1) / n);   // the longer ones first
    cands.push_back(b);
  }
  for (int shortLen = ns; shortLen * 2 < p.Ho; shortLen += std::max(1, p.Ho / 16))   // one long + one short segment
    cands.push_back({0, p.Ho - shortLen, p.Ho});
  for (int s2 = ns; s2 * 4 < p.Ho; s2 += std::max(1, p.Ho / 12))                      // long + medium + short
    for (int s1 = s2 + std::max(1, p.Ho / 12); s1 + s2 < p.Ho - s1; s1 += std::max(1, p.Ho / 12))
      cands.push_back({0, p.Ho - s1 - s2, p.Ho - s2, p.Ho});
#ifdef QCNN_EXPERIMENT     // variant builds only (scripts/build_variant.sh -DQCNN_EXPERIMENT): exactly that many equal segments
  if (const char* e = getenv("QCNN_SLIDE_SEGS")) {
    const int n = std::max(1, std::min(atoi(e), std::min(QK_MAX_SEGS, p.Ho / ns)));
    std::vector<int> b(n + 1);
    for (int i = 0; i <= n; ++i) b[i] = (int)(((long long)p.Ho * i + n - 1) / n);
    cands.assign(1, b);
    tileCost = 1e30;
  }
#endif
  // sliding must beat the (split) tile launch — clearly (8 %) when it needs more channel chunks than the tile kernel: the
  // model does not see the uneven last chunk (VGG-16's 14 x 14 x 512 layers measured 5 % slower where it predicted a tie)
  double best = tileCost * (sc.sl.chunks > qk_conv_slots(Ctg, p.grp).chunks ? 0.92 : 1.0);
  for (const std::vector<int>& b : cands) {
    // order the segments longest first (dispatch order = LPT); boundaries stay contiguous per segment
    std::vector<std::pair<int, int> > segs;
    for (size_t i = 0; i + 1 < b.size(); ++i) segs.push_back({b[i], b[i + 1]});
    std::stable_sort(segs.begin(), segs.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) {
      return x.second - x.first > y.second - y.first; });
    // the kernel reads segment i as [segBeg[i], segBeg[i + 1]): only orders that keep the boundaries monotone fit that
    // encoding — equal cuts and "long, short" do (longest first = left to right)
    bool monotone = true;
    for (size_t i = 0; i + 1 < segs.size(); ++i) monotone = monotone && segs[i].second == segs[i + 1].first;
    if (!monotone || (int)segs.size() > QK_MAX_SEGS) continue;
    const double c = makespan(b);
    if (const char* dbg = getenv("QCNN_DEBUG_PLAN"); dbg && atoi(dbg)) {
      fprintf(stderr, "[qcnn plan] slide Ho=%d Wo=%d panels=%d ny=%d: segs", p.Ho, p.Wo, p.panels, ny);
      for (int v : b) fprintf(stderr, " %d", v);
      fprintf(stderr, " -> %.0f stage-times (tile kernel %.0f)\n", c, tileCost);
    }
    if (c < best) {
      best = c;
      p.nSeg = (int)segs.size();
      for (size_t i = 0; i < b.size(); ++i) p.segBeg[i] = b[i];
    }
  }
  return p.nSeg > 0 ? best : 0.0;
}



// predicted duration (in stage-times of the tile kernel, like QkSplitPlan::cost) of a launch over p.panels panels: tiles
// list-scheduled heaviest first on 256 CUs.  A stage of this kernel is priced by its look-ups: measured (AlexNet conv2 - 5,
// 1000 images, profiles/r4_*) 2540 + 1.97 x (row look-ups per stage) cycles against ~2500 for a stage of the tile kernel,
// whose look-ups run beside its builder waves; `scale` corrects the whole (1.0 = that calibration)
// Z > 1: every tile cut into Z slices of its stage sequence (ConvParams::splitZ) + the reduction of the partial sums (priced as
// qk_conv_plan prices k_conv_sum: Z slabs read, one written at ~4 TB/s behind a launch)
double qk_conv_sym8_cost(const ConvParams& p, const Qk8Config& cf, double scale, int Z) {
  if (!cf.cpw) return 0.0;
  const int TH = cf.th, TW = cf.tw;
  const int tilesX = (p.Wo + TW - 1) / TW, tilesY = (p.Ho + TH - 1) / TH, tiles = tilesX * tilesY;
  std::vector<double> stages((size_t)tiles);
  double total = 0.0;
  for (int r = 0; r < tiles; ++r) {
    int ty, tx;
    tile_of_rank(r, tilesY, tilesX, ty, tx);
    const int ho0 = ty * TH, wo0 = tx * TW;
    const int hoL = std::min(ho0 + TH, p.Ho) - 1, woL = std::min(wo0 + TW, p.Wo) - 1;
    const int rows = std::min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1) - std::max(0, ho0 * p.stride - p.pad) + 1;
    const int cols = std::min(p.W - 1, woL * p.stride - p.pad + p.knl - 1) - std::max(0, wo0 * p.stride - p.pad) + 1;
    stages[r] = (double)std::max(rows, 0) * std::max(cols, 0) * p.M;
    total += stages[r];
  }
  // row look-ups of one group and channel chunk per panel (border-clipped taps x sub-spaces x channels) per built stage
  auto taps = [&](int n, int nIn) {
    long long t = 0;
    for (int o = 0; o < n; ++o) t += std::min(p.knl - 1, nIn - 1 - (o * p.stride - p.pad)) - std::max(0, -(o * p.stride - p.pad)) + 1;
    return (double)t;
  };
  const double perStage = total > 0.0 ? taps(p.Ho, p.H) * taps(p.Wo, p.W) * p.M * std::min(p.Ct / p.grp, 8 * cf.cpw) / total : 0.0;
  const double factor = scale * (2540.0 + 1.97 * perStage) / 2500.0;
  const int ny = p.grp * cf.chunks;
  const long long wgs = (long long)tiles * p.panels * ny;
  if (wgs >= 8 * 256 && Z <= 1) return (factor * total + 10.0 * tiles) * p.panels * ny / 256.0;
  std::priority_queue<double, std::vector<double>, std::greater<double>> q;
  for (int i = 0; i < 256; ++i) q.push(0.0);
  double end = 0.0;
  const int zz = std::max(Z, 1);
  for (int y = 0; y < ny; ++y)
    for (int r = 0; r < tiles; ++r)
      for (int z = 0; z < zz; ++z)
        for (int k = 0; k < p.panels; ++k) {
          const double t = q.top() + factor * stages[r] / zz + 10.0;
          q.pop(); q.push(t);
          end = std::max(end, t);
        }
  if (zz > 1) {
    const double slab = (double)tiles * p.panels * TH * TW * p.Ct * QCNN_PANEL * 4.0;
    end += (slab * (zz + 1.0) / 4.0e6 + slab * zz / 10.0e6 + 5.0) / 1.1;
  }
  return end;
}



// Segments of the sliding form for a launch over p.panels panels (p.nSeg / p.segBeg are filled) and its predicted duration in
// stage-times: the candidates of qk_conv_plan_slide — one to four equal segments per column, or a long and a short one —
// list-scheduled on 256 CUs with this kernel's stage price (qk_conv_sym8_cost).  0: the layer cannot slide.
double qk_conv_sym8_slide_plan(ConvParams& p, const Qk8Config& cf, double scale) {
  p.nSeg = 0;
  if (!cf.cpw || !cf.slide || p.Ho < 2 * cf.th) return 0.0;
  const int ns = cf.th, nc = cf.tw;
  const int colGroups = (p.Wo + nc - 1) / nc;
  const int ny = p.grp * cf.chunks;
  auto segStages = [&](int cgi, int a, int b) {        // strip of output columns [cgi * nc, ..), output rows [a, b)
    const int wA = cgi * nc, wB = std::min(p.Wo, wA + nc) - 1;
    const int cols = std::min(p.W - 1, wB * p.stride - p.pad + p.knl - 1) - std::max(0, wA * p.stride - p.pad) + 1;
    const int rows = std::min(p.H - 1, (b - 1) * p.stride - p.pad + p.knl - 1) - std::max(0, a * p.stride - p.pad) + 1;
    return (double)std::max(rows, 0) * std::max(cols, 0) * p.M;
  };
  auto taps = [&](int n, int nIn) {
    long long t = 0;
    for (int o = 0; o < n; ++o) t += std::min(p.knl - 1, nIn - 1 - (o * p.stride - p.pad)) - std::max(0, -(o * p.stride - p.pad)) + 1;
    return (double)t;
  };
  const double lookups = taps(p.Ho, p.H) * taps(p.Wo, p.W) * p.M * std::min(p.Ct / p.grp, 8 * cf.cpw);   // per group, chunk and panel
  std::vector<std::vector<int> > cands;
  for (int n = 1; n <= 4 && n * ns <= p.Ho; ++n) {
    std::vector<int> b(n + 1);
    for (int i = 0; i <= n; ++i) b[i] = (int)(((long long)p.Ho * i + n - 1) / n);   // the longer ones first
    cands.push_back(b);
  }
  for (int shortLen = ns; shortLen * 2 < p.Ho; shortLen += std::max(1, p.Ho / 16)) cands.push_back({0, p.Ho - shortLen, p.Ho});
#ifdef QCNN_EXPERIMENT     // variant builds only (scripts/build_variant.sh -DQCNN_EXPERIMENT): exactly that many equal segments
  if (const char* e = getenv("QCNN_SYM8_SEGS")) {
    const int n = std::max(1, std::min(atoi(e), std::min(QK_MAX_SEGS, p.Ho / ns)));
    std::vector<int> b(n + 1);
    for (int i = 0; i <= n; ++i) b[i] = (int)(((long long)p.Ho * i + n - 1) / n);
    cands.assign(1, b);
  }
#endif
  double best = 0.0;
  std::vector<double> cu(256);
  for (const std::vector<int>& b : cands) {
    const int nSeg = (int)b.size() - 1;
    if (nSeg > QK_MAX_SEGS) continue;
    double total = 0.0;
    for (int sgi = 0; sgi < nSeg; ++sgi)
      for (int wo = 0; wo < colGroups; ++wo) total += segStages(wo, b[sgi], b[sgi + 1]);
    if (total <= 0.0) continue;
    // a stage's price by its look-ups (2540 + 1.97 x look-ups cycles against 2500 of a tile stage) + the store / restart of
    // the slots at every source row's end
    const double factor = scale * (2540.0 + 1.97 * lookups / total) / 2500.0;
    std::fill(cu.begin(), cu.end(), 0.0);
    std::make_heap(cu.begin(), cu.end(), std::greater<double>());
    for (int y = 0; y < ny; ++y)
      for (int sgi = 0; sgi < nSeg; ++sgi)
        for (int wo = 0; wo < colGroups; ++wo)
          for (int pn = 0; pn < p.panels; ++pn) {
            std::pop_heap(cu.begin(), cu.end(), std::greater<double>());
            const int rows = std::min(p.H - 1, (b[sgi + 1] - 1) * p.stride - p.pad + p.knl - 1) - std::max(0, b[sgi] * p.stride - p.pad) + 1;
            cu.back() += factor * segStages(wo, b[sgi], b[sgi + 1]) + 0.3 * std::max(rows, 0) + 12.0;
            std::push_heap(cu.begin(), cu.end(), std::greater<double>());
          }
    const double c = *std::max_element(cu.begin(), cu.end());
    if (const char* dbg = getenv("QCNN_DEBUG_PLAN"); dbg && atoi(dbg) > 1) {
      fprintf(stderr, "[qcnn plan] sym8 slide Ho=%d Wo=%d panels=%d ny=%d factor %.3f: segs", p.Ho, p.Wo, p.panels, ny, factor);
      for (int v : b) fprintf(stderr, " %d", v);
      fprintf(stderr, " -> %.0f stage-times\n", c);
    }
    if (best == 0.0 || c < best) {
      best = c;
      p.nSeg = nSeg;
      for (size_t i = 0; i < b.size(); ++i) p.segBeg[i] = b[i];
    }
  }
  return best;
}



// predicted duration (in stage-times of the tile kernel, like qk_conv_sym8_cost) of a launch over p.panels panels = 2 p.panels
// half panels: tiles list-scheduled heaviest first on 256 CUs; a half-panel stage is priced fixH + perRead x (ds_read_b128 per
// wave-set stage) cycles against 2500 of a tile stage (calibration: qcnn_half8 notes in LABBOOK.md)
double qk_conv_half8_cost(const ConvParams& p, const QkH8Config& cf, double scale) {
  if (!cf.cpw) return 0.0;
  const int TH = cf.th, TW = cf.tw;
  const int tilesX = (p.Wo + TW - 1) / TW, tilesY = (p.Ho + TH - 1) / TH, tiles = tilesX * tilesY;
  std::vector<double> stages((size_t)tiles);
  double total = 0.0;
  for (int r = 0; r < tiles; ++r) {
    int ty, tx;
    tile_of_rank(r, tilesY, tilesX, ty, tx);
    const int ho0 = ty * TH, wo0 = tx * TW;
    const int hoL = std::min(ho0 + TH, p.Ho) - 1, woL = std::min(wo0 + TW, p.Wo) - 1;
    const int rows = std::min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1) - std::max(0, ho0 * p.stride - p.pad) + 1;
    const int cols = std::min(p.W - 1, woL * p.stride - p.pad + p.knl - 1) - std::max(0, wo0 * p.stride - p.pad) + 1;
    stages[r] = (double)std::max(rows, 0) * std::max(cols, 0) * p.M;
    total += stages[r];
  }
  auto taps = [&](int n, int nIn) {
    long long t = 0;
    for (int o = 0; o < n; ++o) t += std::min(p.knl - 1, nIn - 1 - (o * p.stride - p.pad)) - std::max(0, -(o * p.stride - p.pad)) + 1;
    return (double)t;
  };
  // row look-ups (of 64 images) of one group and channel chunk per half panel and built stage
  const double perStage = total > 0.0 ? taps(p.Ho, p.H) * taps(p.Wo, p.W) * p.M * std::min(p.Ct / p.grp, (8 / cf.ws) * cf.cpw) / total : 0.0;
  // (two wave sets: a stage's valid positions rarely split evenly over the sets, and the slower set holds the barrier)
  const double factor = scale * (cf.ws == 2 ? QK_HALF8_TWO_SETS : 1.0) * (QK_HALF8_FIX + QK_HALF8_PER_ROW * perStage) / 2500.0;
  const int ny = p.grp * cf.chunks;
  const int halves = 2 * p.panels;
  const long long wgs = (long long)tiles * halves * ny;
  if (wgs >= 8 * 256) return (factor * total + 10.0 * tiles) * halves * ny / 256.0;
  std::priority_queue<double, std::vector<double>, std::greater<double>> q;
  for (int i = 0; i < 256; ++i) q.push(0.0);
  double end = 0.0;
  for (int y = 0; y < ny; ++y)
    for (int r = 0; r < tiles; ++r)
      for (int k = 0; k < halves; ++k) {
        const double t = q.top() + factor * stages[r] + 10.0;
        q.pop(); q.push(t);
        end = std::max(end, t);
      }
  return end;
}



// Segments of the sliding form for a launch over p.panels panels (p.nSeg / p.segBeg are filled) and its predicted duration in
// stage-times: one to four equal segments per column, or a long and a short one, list-scheduled on 256 CUs with this kernel's
// stage price.  0: the layer cannot slide.
double qk_conv_half8_slide_plan(ConvParams& p, const QkH8Config& cf, double scale) {
  p.nSeg = 0;
  if (!cf.cpw || !cf.slide || p.Ho < 2 * cf.th) return 0.0;
  const int ns = cf.th, nc = cf.tw;
  const int colGroups = (p.Wo + nc - 1) / nc;
  const int ny = p.grp * cf.chunks;
  const int halves = 2 * p.panels;
  auto segStages = [&](int cgi, int a, int b) {        // strip of output columns [cgi * nc, ..), output rows [a, b)
    const int wA = cgi * nc, wB = std::min(p.Wo, wA + nc) - 1;
    const int cols = std::min(p.W - 1, wB * p.stride - p.pad + p.knl - 1) - std::max(0, wA * p.stride - p.pad) + 1;
    const int rows = std::min(p.H - 1, (b - 1) * p.stride - p.pad + p.knl - 1) - std::max(0, a * p.stride - p.pad) + 1;
    return (double)std::max(rows, 0) * std::max(cols, 0) * p.M;
  };
  auto taps = [&](int n, int nIn) {
    long long t = 0;
    for (int o = 0; o < n; ++o) t += std::min(p.knl - 1, nIn - 1 - (o * p.stride - p.pad)) - std::max(0, -(o * p.stride - p.pad)) + 1;
    return (double)t;
  };
  const double lookups = taps(p.Ho, p.H) * taps(p.Wo, p.W) * p.M * std::min(p.Ct / p.grp, (8 / cf.ws) * cf.cpw);   // per group, chunk and half panel
  std::vector<std::vector<int> > cands;
  for (int n = 1; n <= 4 && n * ns <= p.Ho; ++n) {
    std::vector<int> b(n + 1);
    for (int i = 0; i <= n; ++i) b[i] = (int)(((long long)p.Ho * i + n - 1) / n);   // the longer ones first
    cands.push_back(b);
  }
  for (int shortLen = ns; shortLen * 2 < p.Ho; shortLen += std::max(1, p.Ho / 16)) cands.push_back({0, p.Ho - shortLen, p.Ho});
  double best = 0.0;
  std::vector<double> cu(256);
  for (const std::vector<int>& b : cands) {
    const int nSeg = (int)b.size() - 1;
    if (nSeg > QK_MAX_SEGS) continue;
    double total = 0.0;
    for (int sgi = 0; sgi < nSeg; ++sgi)
      for (int wo = 0; wo < colGroups; ++wo) total += segStages(wo, b[sgi], b[sgi + 1]);
    if (total <= 0.0) continue;
    const double factor = scale * (QK_HALF8_FIX + QK_HALF8_PER_ROW * lookups / total) / 2500.0;
    std::fill(cu.begin(), cu.end(), 0.0);
    std::make_heap(cu.begin(), cu.end(), std::greater<double>());
    for (int y = 0; y < ny; ++y)
      for (int sgi = 0; sgi < nSeg; ++sgi)
        for (int wo = 0; wo < colGroups; ++wo)
          for (int pn = 0; pn < halves; ++pn) {
            std::pop_heap(cu.begin(), cu.end(), std::greater<double>());
            const int rows = std::min(p.H - 1, (b[sgi + 1] - 1) * p.stride - p.pad + p.knl - 1) - std::max(0, b[sgi] * p.stride - p.pad) + 1;
            cu.back() += factor * segStages(wo, b[sgi], b[sgi + 1]) + 0.3 * std::max(rows, 0) + 12.0;
            std::push_heap(cu.begin(), cu.end(), std::greater<double>());
          }
    const double c = *std::max_element(cu.begin(), cu.end());
    if (best == 0.0 || c < best) {
      best = c;
      p.nSeg = nSeg;
      for (size_t i = 0; i < b.size(); ++i) p.segBeg[i] = b[i];
    }
  }
  return best;
}


// ------------------------------------------------------------------------------------------------------------------
// one launch: every eligible family priced, then the decision
// ------------------------------------------------------------------------------------------------------------------
QkConvPlan qk_plan_conv(const ConvParams& p, const QkPlanOptions& o) {
  QkConvPlan pl = {};
  pl.plan = qk_conv_plan(p, o.split ? o.scratchFloats : 0);           // no scratch: only the whole-tile launch is priced
  pl.sym8Z = 1;
  const bool f32 = o.lutMode == 1 && !o.inNchw;                      // the symmetric / eight-wave families: f32 MFMA mode, panel input
  pl.symCost = (o.sym && o.hasSym16 && f32) ? qk_conv_sym_cost(p) : 0.0;
  if (o.sym8 && o.hasSym8 && f32) {
    const Qk8Config c8 = qk_conv_sym8_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K);
    pl.sym8Cost = qk_conv_sym8_cost(p, c8, QK_SYM8_STAGE_FACTOR);
    if (pl.sym8Cost > 0.0 && o.split) {
      // a launch of a few hundred tiles (one GPU's share of a sharded batch): every tile in Z slices of its stage sequence,
      // partial sums reduced by k_conv_sum — taken when predicted 3 % cheaper than the whole tiles
      const int tiles8 = ((p.Wo + c8.tw - 1) / c8.tw) * ((p.Ho + c8.th - 1) / c8.th);
      const long long wgs8 = (long long)tiles8 * p.panels * p.grp * c8.chunks;
      for (int Z = 2; Z <= 6 && wgs8 < 2 * 256; ++Z) {
        if ((size_t)tiles8 * Z * p.panels * c8.th * c8.tw * p.Ct * QCNN_PANEL > o.scratchFloats) break;
        if ((long long)p.M * p.knl * p.knl < 6LL * Z) break;       // a slice keeps a few stages
        const double cz = qk_conv_sym8_cost(p, c8, QK_SYM8_STAGE_FACTOR, Z);
        if (cz < 0.97 * pl.sym8Cost) { pl.sym8Cost = cz; pl.sym8Z = Z; }
      }
    }
  }
  if (o.sym8 && o.hasSym8Slide && f32) {
    ConvParams t = p;
    pl.sym8sCost = qk_conv_sym8_slide_plan(t, qk_conv_sym8_slide_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K, p.knl, p.stride), QK_SYM8_STAGE_FACTOR);
    pl.seg8N = t.nSeg;
    for (int i = 0; i <= t.nSeg && i < 9; ++i) pl.seg8Beg[i] = t.segBeg[i];
  }
  if (o.half8 && o.hasHalf8 && f32)
    pl.half8Cost = qk_conv_half8_cost(p, qk_conv_half8_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K), QK_SYM8_STAGE_FACTOR);
  if (o.half8 && o.hasHalf8Slide && f32) {
    ConvParams t = p;
    pl.half8sCost = qk_conv_half8_slide_plan(t, qk_conv_half8_slide_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K, p.knl, p.stride), QK_SYM8_STAGE_FACTOR);
    pl.segHN = t.nSeg;
    for (int i = 0; i <= t.nSeg && i < 9; ++i) pl.segHBeg[i] = t.segBeg[i];
  }
  if (o.slide && o.hasSlide16) {              // sliding variant where it is predicted to beat the (split) tile kernel
    ConvParams t = p;
    pl.slideCost = qk_conv_plan_slide(t, o.slide >= 2 ? 1e30 : pl.plan.cost);   // 2: whenever the layer is eligible (tests)
    pl.segN = t.nSeg;
    for (int i = 0; i <= t.nSeg && i < 9; ++i) pl.segBeg[i] = t.segBeg[i];
  }
  if (const char* dbg = getenv("QCNN_DEBUG_PLAN"); dbg && atoi(dbg))
    fprintf(stderr, "[qcnn plan] %dx%dx%d -> %dx%dx%d panels %d: tile %.0f (Z %d) | slide %.0f (%d segments) | sym %.0f | sym8 %.0f (Z %d) | sym8 sliding "
                    "%.0f (%d segments) | half-panel %.0f | half-panel sliding %.0f (%d segments) stage-times\n",
            p.H, p.W, p.Cin, p.Ho, p.Wo, p.Ct, p.panels, pl.plan.cost, pl.plan.Z, pl.slideCost, pl.segN, pl.symCost, pl.sym8Cost, pl.sym8Z, pl.sym8sCost,
            pl.seg8N, pl.half8Cost, pl.half8sCost, pl.segHN);
  return pl;
}

QkConvChoice qk_choose_conv(const QkConvPlan& pl, const QkPlanOptions& o) {
  QkConvChoice ch = {};
  ch.family = QK_FAM_TILE; ch.splitFrom = 0; ch.Z = 1;
  const bool f32 = o.lutMode == 1 && !o.inNchw;
  const bool free16 = o.sym < 2 && o.slide < 2;                       // no 16-wave family is forced
  const double slideF = QK_SLIDE8_FACTOR * (o.concurrent ? QK_CONCURRENT_STRIP_FACTOR : 1.0);
  const double halfSlideF = QK_HALF8_SLIDE_FACTOR * (o.concurrent ? QK_CONCURRENT_STRIP_FACTOR : 1.0);
  // the best of the families a candidate is compared with, in stage-times corrected per family (measured us per planner unit):
  // 16-wave symmetric x 1.08, 16-wave and eight-wave sliding forms x QK_SLIDE8_FACTOR
  auto others = [&](bool withSym8s, bool withHalf8) {
    double other = pl.plan.cost;
    if (pl.symCost > 0.0 && 1.08 * pl.symCost < other) other = 1.08 * pl.symCost;
    if (pl.segN > 0 && pl.slideCost > 0.0) other = std::min(other, slideF * pl.slideCost);
    if (pl.sym8Cost > 0.0) other = std::min(other, pl.sym8Cost);
    if (withSym8s && pl.sym8sCost > 0.0 && pl.seg8N > 0) other = std::min(other, slideF * pl.sym8sCost);
    if (withHalf8 && pl.half8Cost > 0.0) other = std::min(other, pl.half8Cost);
    return other;
  };
  auto segs = [&](int n, const int* beg) { ch.nSeg = n; for (int i = 0; i <= n && i < 9; ++i) ch.segBeg[i] = beg[i]; };
  // half-panel eight-wave workgroups, sliding form: forced (QCNN_OPT_HALF8 = 3), or predicted at least 3 % faster than every
  // other plan INCLUDING the half-panel tile form (a strip is a coarser work item: x QK_HALF8_SLIDE_FACTOR)
  if (pl.half8sCost > 0.0 && pl.segHN > 0 && f32 && o.half8 != 2 && (o.half8 >= 3 || (o.sym8 < 2 && free16))) {
    if (o.half8 >= 3 || halfSlideF * pl.half8sCost < 0.97 * others(true, true)) {
      ch.family = QK_FAM_HALF8_SLIDE; segs(pl.segHN, pl.segHBeg);
      return ch;
    }
  }
  // ... tile form: forced (2; 3 where the layer cannot slide), or predicted at least 3 % faster
  if (pl.half8Cost > 0.0 && f32 && (o.half8 >= 2 || (o.sym8 < 2 && free16))) {
    if (o.half8 >= 2 || pl.half8Cost < 0.97 * others(true, false)) { ch.family = QK_FAM_HALF8; return ch; }
  }
  // eight-wave symmetric workgroups, tile or sliding form: when forced (QCNN_OPT_SYM8 = 2: tile form, 3: sliding form where
  // eligible), or predicted at least 3 % faster than every other plan of the launch
  const bool may8 = f32 && (o.sym8 >= 2 || free16) && o.half8 < 2;
  if (may8 && pl.sym8sCost > 0.0 && pl.seg8N > 0 && o.sym8 != 2) {
    // (measured per planner unit, 1000 images: the sliding form 1.04 - 1.11 us — VGG-16's layers, AlexNet conv2 / conv5 —, the tile
    // form 0.89 - 0.93.  With 1.15 the sliding form takes VGG-16's 128- and 256-channel layers, none of AlexNet's: 13 x 13 maps
    // leave too few strips — 208 workgroups for conv4)
    if (o.sym8 >= 3 || slideF * pl.sym8sCost < 0.97 * others(false, false)) {
      ch.family = QK_FAM_SYM8_SLIDE; segs(pl.seg8N, pl.seg8Beg);
      return ch;
    }
  }
  if (pl.sym8Cost > 0.0 && may8) {
    double other = pl.plan.cost;                           // tile kernel, whole or split (in stage-times)
    // (qk_conv_sym_cost prices a 16-wave symmetric stage at 1.09 tile stages; measured 1.18: 2952 against 2508 cycles on AlexNet
    // conv2.  A sliding stage is priced 3 % above a tile stage; measured 3330 against 2500 cycles with 12 channels per wave;
    // AlexNet conv5, which must keep sliding — 0.99 against 1.04 ms —, sits at a cost ratio of 1.157)
    if (pl.symCost > 0.0 && 1.08 * pl.symCost < other) other = 1.08 * pl.symCost;
    if (pl.segN > 0 && pl.slideCost > 0.0) other = std::min(other, slideF * pl.slideCost);
    if (o.sym8 >= 2 || pl.sym8Cost < 0.97 * other) { ch.family = QK_FAM_SYM8; ch.Z = pl.sym8Z; return ch; }
  }
  // 16-wave symmetric workgroups: 128-channel layers that neither slide nor split, when predicted at least 3 % faster
  if (pl.symCost > 0.0 && pl.segN == 0 && pl.plan.Z <= 1 && f32 && (o.sym >= 2 || pl.symCost < 0.97 * pl.plan.cost)) {
    ch.family = QK_FAM_SYM16;
    return ch;
  }
  if (pl.segN > 0) { ch.family = QK_FAM_SLIDE16; segs(pl.segN, pl.segBeg); return ch; }
  if (pl.plan.Z > 1) { ch.splitFrom = pl.plan.splitFrom; ch.Z = pl.plan.Z; }
  return ch;
}

// Diagnostic / test entry (extern "C", plain ints): the plan and the choice for one conv launch.
//   geom[14] = H, W, Cin, Ho, Wo, Ct, knl, stride, pad, grp, M, Cs, K, panels
//   opts[8]  = split, slide, sym, sym8, half8, lutMode, flags (1: NCHW input read in place, 2: concurrent sub-batches), scratch (in Mi floats)
// The program tables a layer would have are derived from its shape exactly as qcnn_model_commit plans them.
//   costs[7] = tile, 16-wave sliding, 16-wave symmetric, eight-wave tile, eight-wave sliding, half-panel tile, half-panel sliding
//   choice[13] = family, splitFrom, Z, nSeg, segBeg[0..8]
extern "C" int qcnn_plan_conv_query(const int* geom, const int* opts, double* costs, int* choice) {
  if (!geom || !opts) return 1;
  ConvParams p = {};
  p.H = geom[0]; p.W = geom[1]; p.Cin = geom[2]; p.Ho = geom[3]; p.Wo = geom[4]; p.Ct = geom[5];
  p.knl = geom[6]; p.stride = geom[7]; p.pad = geom[8]; p.grp = geom[9]; p.M = geom[10]; p.Cs = geom[11]; p.K = geom[12]; p.panels = geom[13];
  p.pd = 1; p.splitZ = 1;
  if (p.grp < 1 || p.Ct % p.grp || p.Cin % p.grp || p.panels < 1 || p.knl < 1 || p.stride < 1) return 1;
  QkPlanOptions o = {};
  o.split = opts[0]; o.slide = opts[1]; o.sym = opts[2]; o.sym8 = opts[3]; o.half8 = opts[4]; o.lutMode = opts[5]; o.inNchw = opts[6] & 1; o.concurrent = (opts[6] >> 1) & 1;
  o.scratchFloats = (size_t)opts[7] << 20;
  const int Ctg = p.Ct / p.grp;
  o.hasSlide16 = p.K == 128 && qk_slide_config(Ctg, p.grp, p.knl, p.stride).ns > 0;
  o.hasSym16 = qk_conv_sym_shape(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K);
  o.hasSym8 = qk_conv_sym8_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K).cpw != 0;
  o.hasSym8Slide = qk_conv_sym8_slide_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K, p.knl, p.stride).cpw != 0;
  o.hasHalf8 = qk_conv_half8_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K).cpw != 0;
  o.hasHalf8Slide = qk_conv_half8_slide_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K, p.knl, p.stride).cpw != 0;
  const QkConvPlan pl = qk_plan_conv(p, o);
  const QkConvChoice ch = qk_choose_conv(pl, o);
  if (costs) {
    costs[0] = pl.plan.cost; costs[1] = pl.slideCost; costs[2] = pl.symCost; costs[3] = pl.sym8Cost; costs[4] = pl.sym8sCost;
    costs[5] = pl.half8Cost; costs[6] = pl.half8sCost;
  }
  if (choice) {
    choice[0] = ch.family; choice[1] = ch.splitFrom; choice[2] = ch.Z; choice[3] = ch.nSeg;
    for (int i = 0; i < 9; ++i) choice[4 + i] = ch.segBeg[i];
  }
  return 0;
}
