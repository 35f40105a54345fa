// qcnn_engine.hip — the C-ABI of include/qcnn_hip.h: context, model planning, layer loop.
//
// Mirrors the control flow of the reference's CaffeEva (src/CaffeEva.cc): LoadCaffePara ->
// PrepFeatMap/PrepFeatBuf/PrepCtrdBuf/PrepAsmtBuf (:109-149) becomes qcnn_model_begin / commit /
// set_layer_params; ExecForwardPass's layer loop (:184-205, :232-254) becomes run_layers().
// Everything the loop touches lives in HBM for the whole batch; the host only enqueues kernels.
#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/qcnn_hip.h"
#include "qcnn_kernels.h"
#include "qcnn_planner.h"

namespace {

std::string g_createError = "";

struct LayerShape {
  int M = 0, K = 0, Cs = 0;                                    // as the kernels see the layer (K <= 128)
  int P = 1, Mfile = 0, Kfile = 0;                             // a parameter set with 128 < K <= 256 code words per sub-space: every sub-space is P =
                                                               // ceil(K / 127) pseudo sub-spaces of <= 127 code words + a zero row (M = P * Mfile, K = 128)
  bool dense = false;                                          // precise path: bias + dense weights instead of a quantisation
  size_t offDense = 0, denseFloats = 0;
  size_t offBias = 0, offCtrd = 0, offAsmt = 0, offDmap = 0;   // byte offsets into the arena
  size_t asmtBytes = 0;
  size_t offProg = 0, progBytes = 0;                           // conv with K = 128: offsets in consumption order (QkProgram)
  size_t offProgS = 0, progSBytes = 0;                         // ... and in the order of the sliding variant, where it applies
  size_t offProgY = 0, progYBytes = 0;                         // ... and of the symmetric kernel's (8 channels per wave, 2x2 tile) layout
  size_t offProg8 = 0, prog8Bytes = 0;                         // ... and of the eight-wave symmetric kernel's layout (Qk8Config)
  size_t offProg8S = 0, prog8SBytes = 0;                       // ... and of its sliding form (qk_conv_sym8_slide_config)
  size_t offProgH8 = 0, progH8Bytes = 0;                       // ... and of the half-panel eight-wave kernel (QkH8Config, qcnn_half8.hip)
  size_t offProgH8S = 0, progH8SBytes = 0;                     // ... and of its sliding form (qk_conv_half8_slide_config)
  size_t offCtrd8 = 0;                                         // ... with the code book in that kernel's operand order (qk_ctrd8_index)
  size_t offProgF8 = 0, progF8Bytes = 0, offCtrdF = 0;         // FC with 32 code words of 4 dims: program + code book of the eight-wave kernel (k_fc_sym8)
  size_t offCbn = 0, cbnBytes = 0; int cbnBits = 0;            // FC: the assignments bit-packed as the .cbn payload holds them (file order
                                                               // [Ct][M], include/FileIO.h:128-166), read in place by the few-image kernel
  size_t offDecN = 0; int decNV = 0;                          // first layer: the same code words in k_conv_dec_nchw's order; decNV: its padded k (0: not eligible)
  size_t offDec = 0; int decKp = 0, decS = 0;                  // decoded code words (qcnn_decoded.hip): conv layer with one sub-space of
                                                               // <= 4 dims (decKp > 0), FC layer with one-dim sub-spaces (decKp = -1); 0: not eligible
  bool hasDmap = false;
  bool loaded = false;
  // QCNN_OPT_LUT_MODE = 2 (fp16 table storage): the eight-wave kernels' program tables with offsets into the fp16 table layout.
  // Own allocations, built from the arena's assignment bytes when the mode first runs the layer (every rank of a group builds
  // its own from the broadcast arena); dropped when the layer's parameters are uploaded again
  uint16_t* prog8H = nullptr;
  uint16_t* progF8H = nullptr;
  uint16_t* prog8A = nullptr;        // QCNN_OPT_LUT_MODE = 3 (fp16 sums too): the program of the twice-as-large tiles (qk_conv_sym8_config16)
  // conv: launch plans (qcnn_planner.h) by launch geometry and options (panels, sub-batches, split / slide / sym, LUT mode, input
  // in place): sub-batches of unequal panel counts (3 panels over 2 streams) each keep theirs instead of evicting one another —
  // a plan is dozens of 256-CU list schedules on the host
  std::map<long long, QkConvPlan> plans;
  int segN = 0, segBeg[9] = {0};                               // segments of the last launch when it slid (qcnn_get_layer_segments)
  int lastFrom = -1, lastZ = 1;                                // how the last launch was actually cut
};

struct FmDims { int h, w, c; };

constexpr int kProfRing = 32;    // forwards whose per-layer events are kept
constexpr int kMaxStreams = 4;   // sub-batches (streams) of one forward
constexpr int kSmallBatchMax = QCNN_SMALL_BATCH_MAX;  // batches up to this size run the few-image kernels (QCNN_OPT_SMALL_BATCH): beyond, a
                                   // 128-image panel is cheaper (measured: 1 / 2 / 3 / 4 images 0.58 / 0.85 / 1.15 / 1.47 ms, a panel 1.50 ms)
constexpr int kMaxFcSplit = 32;  // workgroups along the sub-space axis of an FC layer (partial sums reduced in fixed order)
constexpr size_t kConvPartialFloats = (size_t)64 << 20;   // 256 MB of partial sums for split conv tiles (all sub-batches), allocated when a plan first splits
constexpr size_t kSlack = 64 * 1024;   // bytes of slack behind every device buffer: the MFMA operand loads are
                                        // unconditional and may read a few rows past the last dim / sub-space

}  // namespace

struct QcnnCtx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool ownStream = false;
  std::string err;
  int lutMode = 1, keepAll = 1, profile = 0;
  int nStreams = 2;                  // QCNN_OPT_STREAMS: sub-batches of whole panels run concurrently
  int smallBatch = 1;                // QCNN_OPT_SMALL_BATCH: few-image kernels for batches <= kSmallBatchMax
  int hostChunk = 2;                 // QCNN_OPT_HOST_CHUNK: panels per chunk of a large qcnn_forward_host batch (0: one launch)
  int directDec = 1;                 // QCNN_OPT_DIRECT_DEC: a decoded first layer reads the NCHW input in place (k_conv_dec_nchw) on the fast path
  int packedFc = 0;                  // QCNN_OPT_PACKED_FC (default off: measured 0.056 against 0.035 ms for AlexNet fc6 at one image): the few-image FC kernel reads the bit-packed assignment stream in place
  int half8 = 1;                     // QCNN_OPT_HALF8: half-panel eight-wave workgroups where predicted faster (2: whenever eligible)
  int sym8 = 1;                      // QCNN_OPT_SYM8: eight-wave symmetric workgroups where predicted faster (2: whenever eligible)
  int sym = 1;                       // QCNN_OPT_SYM: symmetric workgroups for 128-channel layers where predicted faster (2: whenever eligible)
  int decode = 1;                    // QCNN_OPT_DECODE: one-sub-space conv layers through their decoded code words (MFMA builders only)
  int slide = 1;                     // QCNN_OPT_SLIDE: sliding-window conv kernels where they pay (MFMA builders only)
  int split = 1;                     // QCNN_OPT_SPLIT: launches that do not fill the chip split their tail (MFMA builders only)
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};
  hipEvent_t evFork = nullptr, evJoin[3] = {nullptr, nullptr, nullptr};

  int L = 0, inC = 0, inH = 0, inW = 0;
  std::vector<QcnnLayerDesc> layers;
  std::vector<FmDims> dims;          // L + 1
  std::vector<LayerShape> shapes;    // L
  int firstFc = -1;
  bool committed = false;
  int maxBatch = 0, maxPanels = 0;

  char* arena = nullptr;
  bool ownArena = false;
  size_t arenaBytes = 0;
  std::vector<float*> fmBuf;         // L + 1, own allocations (nullptr where aliased)
  float* stageIn = nullptr;          // [maxBatch][maxE] linear staging for host <-> device conversions
  float* stageOut = nullptr;
  size_t stageElems = 0;
  uint16_t* stageTop5 = nullptr;
  // pipelined host path (qcnn_forward_host_batches): the upload of batch b+1 runs on its own stream under the layers of
  // batch b; two device input buffers (stageIn and stageIn1), two pinned host result buffers
  hipStream_t copyStream = nullptr;
  float* stageIn1 = nullptr;         // [maxBatch][inC*inH*inW]
  float* pinProb[2] = {nullptr, nullptr};
  uint16_t* pinTop5[2] = {nullptr, nullptr};
  hipEvent_t evCopied[2] = {nullptr, nullptr}, evFreed[2] = {nullptr, nullptr}, evDone[2] = {nullptr, nullptr};
  bool freedValid[2] = {false, false};
  std::vector<hipEvent_t> evChunk;   // chunked single batch (qcnn_forward_host): "chunk k is on the device"
  float* fcFlat = nullptr;           // first FC layer's input in consumption order
  float* fcPartial = nullptr;        // split-M partial sums of the FC layers
  float* convPartial = nullptr;      // partial sums of split conv tiles (kConvPartialFloats)
  bool noConvPartial = false;        // ... could not be allocated: split plans launch their tiles whole
  size_t fcPartialElems = 0;
  size_t fcMaxCt = 0;
  int lastN = 0;
  std::vector<float*> lastFm;        // pointer table of the last forward

  std::vector<hipEvent_t> ev;        // kProfRing * kMaxStreams * L * 2
  int profCount = 0;                 // forwards in the ring, not yet drained
  struct ProfRec { size_t slot; int layer; };
  std::vector<ProfRec> profPending;  // event pairs recorded since the last drain (only layers that were launched)
  std::vector<double> profSum;       // per layer: ms summed over every recorded launch
  std::vector<long long> profLaunches;
  int profForwards = 0;
};

namespace {

int fail(QcnnCtx* c, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_createError = buf;
  return 1;
}

#define HIP_TRY(c, call)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) return fail((c), "%s -> %s", #call, hipGetErrorString(e_));     \
  } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int conv_out(int in, int knl, int stride, int pad) { return (in + 2 * pad - knl) / stride + 1; }
int pool_out(int in, int knl, int stride, int pad) {          // ceil mode, src/CaffeEva.cc:367-370
  const int num = in + 2 * pad - knl;
  return (num + stride - 1) / stride + 1;
}

size_t fm_elems(const QcnnCtx* c, int l) { return (size_t)c->dims[l].h * c->dims[l].w * c->dims[l].c; }

int plan_arena(QcnnCtx* c) {
  size_t off = 0;
  for (int l = 0; l < c->L; ++l) {
    const QcnnLayerDesc& d = c->layers[l];
    if (d.type != QCNN_CONV && d.type != QCNN_FCNT) continue;
    LayerShape& s = c->shapes[l];
    const int Ct = c->dims[l + 1].c;
    if (s.dense) {                     // precise path: bias + weights [grp][taps][Cin/grp][Ct/grp]
      s.offBias = off; off = align_up(off + sizeof(float) * Ct, 256);
      const size_t taps = (d.type == QCNN_CONV) ? (size_t)d.knlSiz * d.knlSiz : 1;
      const size_t cin = (d.type == QCNN_CONV) ? (size_t)c->dims[l].c / d.grpCnt : fm_elems(c, l);
      s.denseFloats = taps * cin * Ct;
      s.offDense = off; off = align_up(off + sizeof(float) * s.denseFloats, 256);
      s.hasDmap = (d.type == QCNN_FCNT && l == c->firstFc && c->dims[l].h * c->dims[l].w > 1);
      if (s.hasDmap) { s.offDmap = off; off = align_up(off + sizeof(int) * fm_elems(c, l), 256); }
      continue;
    }
    if (s.K <= 0) return fail(c, "layer %d: neither a quantisation shape (qcnn_model_set_layer_shape) nor dense weights (qcnn_model_set_layer_dense) declared", l);
    s.offBias = off; off = align_up(off + sizeof(float) * Ct, 256);
    s.offCtrd = off; off = align_up(off + sizeof(float) * (size_t)s.M * s.Cs * s.K, 256);
    // assignment table: one-byte row slots in the order the gather waves consume them (QkSlots, qcnn_kernels.h)
    const QkSlots sl = (d.type == QCNN_CONV) ? qk_conv_slots(Ct / d.grpCnt, d.grpCnt) : qk_fc_slots(Ct);
    const size_t taps = (d.type == QCNN_CONV) ? (size_t)d.knlSiz * d.knlSiz : 1;
    s.asmtBytes = taps * s.M * sl.rowStride;
    s.offAsmt = off; off = align_up(off + s.asmtBytes + QCNN_ROWS_PAD, 256);
    s.progBytes = 0;
    // (layers of pseudo sub-spaces — more than 128 code words, s.P > 1 — always run the exact-builder kernel, which reads the plain
    // table: none of the program tables / operand-order code books below is built for them)
    if (d.type == QCNN_CONV && s.K == 128 && s.P == 1) {     // the MFMA panel kernel reads its offsets through the program table
      const QkProgram pg = qk_conv_program(sl, d.knlSiz, d.stride);
      s.progBytes = (size_t)pg.rfH * pg.rfW * s.M * pg.rowU16 * sizeof(uint16_t);
      s.offProg = off; off = align_up(off + s.progBytes + QCNN_ROWS_PAD, 256);
      s.progSBytes = 0;
      const QkSlide sc = qk_slide_config(Ct / d.grpCnt, d.grpCnt, d.knlSiz, d.stride);
      if (sc.ns > 0) {
        const QkProgram ps = qk_conv_program_slide(sc.sl, sc.ns, sc.nc, d.knlSiz, d.stride);
        s.progSBytes = (size_t)ps.rfH * ps.rfW * s.M * ps.rowU16 * sizeof(uint16_t);
        s.offProgS = off; off = align_up(off + s.progSBytes + QCNN_ROWS_PAD, 256);
      }
    }
    s.progYBytes = 0;
    if (d.type == QCNN_CONV && s.P == 1 && qk_conv_sym_shape(c->dims[l].c, d.grpCnt, Ct, s.M, s.Cs, s.K)) {
      const QkProgram py = qk_conv_program(qk_make_slots(Ct / d.grpCnt, d.grpCnt, 8), d.knlSiz, d.stride);
      s.progYBytes = (size_t)py.rfH * py.rfW * s.M * py.rowU16 * sizeof(uint16_t);
      s.offProgY = off; off = align_up(off + s.progYBytes + QCNN_ROWS_PAD, 256);
    }
    s.progF8Bytes = 0;
    if (d.type == QCNN_FCNT && s.P == 1 && qk_fc_sym8_shape((int)fm_elems(c, l), Ct, s.M, s.Cs, s.K)) {
      s.progF8Bytes = qk_fc_sym8_program_bytes(Ct, s.M);
      s.offProgF8 = off; off = align_up(off + s.progF8Bytes + QCNN_ROWS_PAD, 256);
      s.offCtrdF = off; off = align_up(off + sizeof(float) * (size_t)s.M * s.Cs * s.K, 256);
    }
    s.cbnBytes = 0;
    if (d.type == QCNN_FCNT && s.P == 1) {           // bits = the reference's CalcBitCntPerEle for K code words (src/CaffePara.cc:360-380)
      s.cbnBits = 1;
      while ((1 << s.cbnBits) < s.K) ++s.cbnBits;
      const size_t per = 4096 * 8 / (size_t)s.cbnBits;
      s.cbnBytes = ((size_t)Ct * s.M + per - 1) / per * 4096;
      s.offCbn = off; off = align_up(off + s.cbnBytes + 256, 256);
    }
    s.prog8Bytes = 0; s.prog8SBytes = 0; s.progH8Bytes = 0; s.progH8SBytes = 0;
    if (d.type == QCNN_CONV && s.P == 1) {
      const Qk8Config c8 = qk_conv_sym8_config(c->dims[l].c, d.grpCnt, Ct, s.M, s.Cs, s.K);
      s.prog8Bytes = qk_conv_sym8_program_bytes(c8, d.grpCnt, d.knlSiz, d.stride, s.M);
      const Qk8Config c8s = qk_conv_sym8_slide_config(c->dims[l].c, d.grpCnt, Ct, s.M, s.Cs, s.K, d.knlSiz, d.stride);
      s.prog8SBytes = qk_conv_sym8_program_bytes(c8s, d.grpCnt, d.knlSiz, d.stride, s.M);
      const QkH8Config ch8 = qk_conv_half8_config(c->dims[l].c, d.grpCnt, Ct, s.M, s.Cs, s.K);
      s.progH8Bytes = qk_conv_half8_program_bytes(ch8, d.grpCnt, d.knlSiz, d.stride, s.M);
      if (s.progH8Bytes) { s.offProgH8 = off; off = align_up(off + s.progH8Bytes + QCNN_ROWS_PAD, 256); }
      const QkH8Config ch8s = qk_conv_half8_slide_config(c->dims[l].c, d.grpCnt, Ct, s.M, s.Cs, s.K, d.knlSiz, d.stride);
      s.progH8SBytes = qk_conv_half8_program_bytes(ch8s, d.grpCnt, d.knlSiz, d.stride, s.M);
      if (s.progH8SBytes) { s.offProgH8S = off; off = align_up(off + s.progH8SBytes + QCNN_ROWS_PAD, 256); }
      if (s.prog8Bytes) { s.offProg8 = off; off = align_up(off + s.prog8Bytes + QCNN_ROWS_PAD, 256); }
      if (s.prog8SBytes) { s.offProg8S = off; off = align_up(off + s.prog8SBytes + QCNN_ROWS_PAD, 256); }
      if (s.prog8Bytes || s.prog8SBytes || s.progH8Bytes) { s.offCtrd8 = off; off = align_up(off + sizeof(float) * (size_t)s.M * s.Cs * s.K, 256); }
    }
    s.decKp = 0;
    s.hasDmap = (d.type == QCNN_FCNT && l == c->firstFc && c->dims[l].h * c->dims[l].w > 1);
    if (d.type == QCNN_CONV && qk_conv_dec_shape(c->dims[l].c, d.grpCnt, s.M, Ct, d.knlSiz, &s.decKp, &s.decS)) {
      s.offDec = off; off = align_up(off + sizeof(float) * (size_t)d.knlSiz * s.decKp * s.decS, 256);
      int nS = 0;
      s.decNV = 0;
      if (l == 0 && qk_conv_dec_nchw_shape(c->dims[l].c, d.grpCnt, s.M, Ct, d.knlSiz, d.padSiz, &s.decNV, &nS)) {
        s.offDecN = off; off = align_up(off + sizeof(float) * (size_t)s.decNV * nS, 256);
      } else {
        s.decNV = 0;
      }
    } else if (d.type == QCNN_FCNT && !s.hasDmap && qk_fc_dec_shape((int)fm_elems(c, l), s.M, s.Cs, Ct, &s.decS)) {
      s.decKp = -1;                   // FC layer with one-dim sub-spaces: [D][decS] decoded code words
      s.offDec = off; off = align_up(off + sizeof(float) * fm_elems(c, l) * s.decS, 256);
    } else {
      s.decKp = 0;
    }
    if (s.hasDmap) { s.offDmap = off; off = align_up(off + sizeof(int) * fm_elems(c, l), 256); }
  }
  c->arenaBytes = off + kSlack;
  return 0;
}

void drop_f16_programs(LayerShape& s) {
  if (s.prog8H) (void)hipFree(s.prog8H);
  if (s.progF8H) (void)hipFree(s.progF8H);
  if (s.prog8A) (void)hipFree(s.prog8A);
  s.prog8H = nullptr; s.progF8H = nullptr; s.prog8A = nullptr;
}

void free_model(QcnnCtx* c) {
  for (LayerShape& s : c->shapes) drop_f16_programs(s);
  for (float* p : c->fmBuf) if (p) (void)hipFree(p);
  c->fmBuf.clear();
  if (c->ownArena && c->arena) (void)hipFree(c->arena);
  c->arena = nullptr; c->ownArena = false;
  if (c->stageIn) (void)hipFree(c->stageIn);
  if (c->stageOut) (void)hipFree(c->stageOut);
  if (c->stageTop5) (void)hipFree(c->stageTop5);
  if (c->stageIn1) (void)hipFree(c->stageIn1);
  c->stageIn1 = nullptr;
  for (int k = 0; k < 2; ++k) {
    if (c->pinProb[k]) (void)hipHostFree(c->pinProb[k]);
    if (c->pinTop5[k]) (void)hipHostFree(c->pinTop5[k]);
    c->pinProb[k] = nullptr; c->pinTop5[k] = nullptr; c->freedValid[k] = false;
  }
  if (c->fcPartial) (void)hipFree(c->fcPartial);
  if (c->convPartial) (void)hipFree(c->convPartial);
  c->convPartial = nullptr; c->noConvPartial = false;
  if (c->fcFlat) (void)hipFree(c->fcFlat);
  c->fcFlat = nullptr;
  c->stageIn = c->stageOut = nullptr; c->stageTop5 = nullptr; c->stageElems = 0;
  c->fcPartial = nullptr; c->fcPartialElems = 0;
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  c->ev.clear();
  c->profCount = 0; c->profPending.clear(); c->profForwards = 0;
  c->lastFm.clear(); c->lastN = 0;       // nothing of the old model can be read back any more
  c->committed = false;
}

// Linear staging for host <-> device conversions.  Sized for what the forward paths need — a batch of network inputs, a batch of
// class scores — and grown on demand when a larger feature map is dumped (qcnn_get_layer_output / qcnn_run_layer): sizing it for the
// LARGEST map of the model up front made the first qcnn_forward_host of VGG-16 at batch 1000 allocate 2 x 12.8 GB (0.75 s; AlexNet: 2 x
// 1.2 GB)
int ensure_stage(QcnnCtx* c, size_t elems = 0) {
  const size_t base = std::max(fm_elems(c, 0), fm_elems(c, c->L)) * (size_t)c->maxBatch;
  const size_t need = std::max(base, elems);
  if (c->stageIn && c->stageElems >= need) return 0;
  if (c->stageIn) {                                     // grow: nothing may still be using the old buffers
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->copyStream) HIP_TRY(c, hipStreamSynchronize(c->copyStream));
    (void)hipFree(c->stageIn); (void)hipFree(c->stageOut);
    c->stageIn = c->stageOut = nullptr;
  }
  c->stageElems = 0;                                    // committed only when BOTH buffers exist (a half-grown pair must never be used)
  float *in = nullptr, *out = nullptr;
  HIP_TRY(c, hipMalloc(&in, need * sizeof(float) + kSlack));
  if (hipMalloc(&out, need * sizeof(float) + kSlack) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(in);
    return fail(c, "staging buffers: 2 x %zu bytes do not fit the device", need * sizeof(float) + kSlack);
  }
  c->stageIn = in; c->stageOut = out; c->stageElems = need;
  if (!c->stageTop5) HIP_TRY(c, hipMalloc(&c->stageTop5, (size_t)c->maxBatch * 5 * sizeof(uint16_t)));
  return 0;
}

// buffers, stream and events of the pipelined host path
int ensure_pipeline(QcnnCtx* c) {
  if (ensure_stage(c)) return 1;
  if (c->stageIn1) return 0;
  const size_t inElems = fm_elems(c, 0) * c->maxBatch;
  const size_t classes = fm_elems(c, c->L);
  for (int k = 0; k < 2; ++k) {
    if (!c->evCopied[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->evCopied[k], hipEventDisableTiming));
    if (!c->evFreed[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->evFreed[k], hipEventDisableTiming));
    if (!c->evDone[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->evDone[k], hipEventDisableTiming));
    HIP_TRY(c, hipHostMalloc(&c->pinProb[k], (size_t)c->maxBatch * classes * sizeof(float), hipHostMallocPortable));
    HIP_TRY(c, hipHostMalloc(&c->pinTop5[k], (size_t)c->maxBatch * 5 * sizeof(uint16_t), hipHostMallocPortable));
    c->freedValid[k] = false;
  }
  HIP_TRY(c, hipMalloc(&c->stageIn1, inElems * sizeof(float) + kSlack));
  return 0;
}

// One layer on `panels` panels: src/dst in panel layout.  flatFcInput: the FC input rows are already
// in consumption order (qcnn_run_layer), so the NCHW-flatten map is not applied.
// p0: first panel of the sub-batch (offsets into the scratch buffers), st: the stream it runs on
// live: images every panel of this launch holds (128, or the batch size of a single-panel forward); small: the
// few-image kernels (qcnn_small.hip) run the conv/FC layers
// sub / nsub: index and number of the sub-batches (streams) of this forward: each has its own share of the scratch
// Does conv layer l run through its decoded code words (qcnn_decoded.hip)?  The f32 MFMA mode only: the exact
// builder keeps the reference's summation order and the fp16 study is about the tables themselves.
bool decoded_layer(const QcnnCtx* c, int l) {
  const LayerShape& s = c->shapes[l];
  return c->decode && s.decKp > 0 && !s.dense && c->lutMode == 1 && c->layers[l].type == QCNN_CONV;
}
bool decoded_fc(const QcnnCtx* c, int l) {
  const LayerShape& s = c->shapes[l];
  return c->decode && s.decKp < 0 && !s.dense && c->lutMode == 1 && c->layers[l].type == QCNN_FCNT;
}

// fp16 table storage: the layer's program table in the fp16 layout's offsets (first use; the build runs on `st`, in front of the
// launch that reads it)
// a table pointer is published in the LayerShape only once its build has been enqueued without error: on any failure behind the
// hipMalloc the allocation is released, so that a later forward builds again instead of launching with an unbuilt table
#define F16_TRY(ptr, expr)                                                              \
  do {                                                                                  \
    const hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                             \
      (void)hipFree(ptr);                                                               \
      return fail(c, "%s -> %s", #expr, hipGetErrorString(e_));                         \
    }                                                                                   \
  } while (0)
int ensure_f16_program(QcnnCtx* c, int l, hipStream_t st) {
  const QcnnLayerDesc& d = c->layers[l];
  LayerShape& s = c->shapes[l];
  const int Ct = c->dims[l + 1].c;
  bool built = false;
  if (d.type == QCNN_CONV && s.prog8Bytes && c->lutMode == 3 && !s.prog8A) {
    built = true;
    const Qk8Config cf = qk_conv_sym8_config16(c->dims[l].c, d.grpCnt, Ct, s.M, s.Cs, s.K);
    const size_t bytes = qk_conv_sym8_program_bytes(cf, d.grpCnt, d.knlSiz, d.stride, s.M);
    uint16_t* t = nullptr;
    HIP_TRY(c, hipMalloc(&t, bytes + QCNN_ROWS_PAD + 4096));
    F16_TRY(t, hipMemsetAsync(t, 0, bytes + QCNN_ROWS_PAD + 4096, st));
    F16_TRY(t, qk_build_program8(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), t, qk_conv_slots(Ct / d.grpCnt, d.grpCnt),
                                 cf, Ct / d.grpCnt, d.grpCnt, d.knlSiz, d.stride, s.M, st, 1));
    s.prog8A = t;
  }
  if (d.type == QCNN_CONV && s.prog8Bytes && c->lutMode == 2 && !s.prog8H) {
    built = true;
    uint16_t* t = nullptr;
    HIP_TRY(c, hipMalloc(&t, s.prog8Bytes + QCNN_ROWS_PAD));
    F16_TRY(t, hipMemsetAsync(t, 0, s.prog8Bytes + QCNN_ROWS_PAD, st));
    F16_TRY(t, qk_build_program8(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), t, qk_conv_slots(Ct / d.grpCnt, d.grpCnt),
                                 qk_conv_sym8_config(c->dims[l].c, d.grpCnt, Ct, s.M, s.Cs, s.K), Ct / d.grpCnt, d.grpCnt, d.knlSiz,
                                 d.stride, s.M, st, 1));
    s.prog8H = t;
  }
  if (d.type == QCNN_FCNT && s.progF8Bytes && !s.progF8H) {
    uint16_t* t = nullptr;
    HIP_TRY(c, hipMalloc(&t, s.progF8Bytes + QCNN_ROWS_PAD));
    F16_TRY(t, hipMemsetAsync(t, 0, s.progF8Bytes + QCNN_ROWS_PAD, st));
    F16_TRY(t, qk_build_program_fc8(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), t, qk_fc_slots(Ct), Ct, s.M, st, 1));
    s.progF8H = t;
  } else if (!built) {
    return 0;
  }
  HIP_TRY(c, hipStreamSynchronize(st));       // once per layer: the other sub-batch streams of this forward read the table too
  return 0;
}

int launch_layer(QcnnCtx* c, int l, const float* src, float* dst, int panels, bool fuseRelu, bool flatFcInput,
                 int p0, hipStream_t st, const float* inNchw = nullptr, int nImages = 0, int live = QCNN_PANEL,
                 bool small = false, int sub = 0, int nsub = 1, int panelsAll = 0) {
  // panelsAll: panels of ALL sub-batches of this forward (they run concurrently on their own streams and share the 256 CUs); 0 = this
  // launch is alone
  const QcnnLayerDesc& d = c->layers[l];
  const FmDims& a = c->dims[l];
  const FmDims& b = c->dims[l + 1];
  LayerShape& s = c->shapes[l];
  hipError_t e = hipSuccess;
  switch (d.type) {
    case QCNN_CONV: {
      if (!s.loaded) return fail(c, "layer %d: parameters not uploaded", l);
      if (s.dense) {                       // precise path (CalcFeatMap_ConvPrec, src/CaffeEva.cc:681-758)
        DenseParams q;
        q.src = src; q.dst = dst;
        q.bias = reinterpret_cast<const float*>(c->arena + s.offBias);
        q.wt = reinterpret_cast<const float*>(c->arena + s.offDense);
        q.H = a.h; q.W = a.w; q.Cin = a.c; q.Ho = b.h; q.Wo = b.w; q.Ct = b.c;
        q.knl = d.knlSiz; q.stride = d.stride; q.pad = d.padSiz; q.grp = d.grpCnt;
        q.relu = fuseRelu ? 1 : 0; q.panels = panels;
        e = qk_dense(q, st);
        break;
      }
      // one sub-space of <= 4 dims: decoded code words on the matrix pipe.  Batches of one to three images too when the layer
      // reads the NCHW input in place: a 16-image tile with one live image still beats the few-image table kernel, whose
      // workgroups rebuild the pixel tables five times (AlexNet conv1 at one image: 0.126 -> 0.0xx ms)
      if (decoded_layer(c, l) && (small ? (inNchw && s.decNV && c->directDec) : (!inNchw || s.decNV))) {
        DecParams q;
        q.src = src; q.dst = dst;
        q.srcNchw = 0; q.nImages = 0; q.panel0 = 0;
        if (inNchw) { q.src = inNchw; q.srcNchw = 1; q.nImages = nImages; q.panel0 = p0; }   // network input read in place
        q.bias = reinterpret_cast<const float*>(c->arena + s.offBias);
        q.wdec = reinterpret_cast<const float*>(c->arena + (inNchw ? s.offDecN : s.offDec));
        q.H = a.h; q.W = a.w; q.Cin = a.c; q.Ho = b.h; q.Wo = b.w; q.Ct = b.c;
        q.knl = d.knlSiz; q.stride = d.stride; q.pad = d.padSiz;
        q.Kr = d.knlSiz * a.c; q.Kp = s.decKp; q.S = s.decS;
        if (inNchw) { q.Kr = d.knlSiz * d.knlSiz * a.c; q.Kp = s.decNV; q.S = b.c; }
        q.relu = fuseRelu ? 1 : 0; q.panels = panels; q.live = live;
        s.lastFrom = -3; s.lastZ = inNchw ? 2 : 1;        // reported by qcnn_get_layer_split as (-3, 1), NCHW in place: (-3, 2)
        e = inNchw ? qk_conv_dec_nchw(q, st) : qk_conv_dec(q, st);
        if (e != hipErrorInvalidValue) break;             // (a map beyond the kernel's 32-bit byte offsets: the table kernel below)
      }
      ConvParams p;
      p.src = src; p.dst = dst;
      p.srcNchw = 0; p.nImages = 0; p.panel0 = 0;
      if (inNchw) { p.src = inNchw; p.srcNchw = 1; p.nImages = nImages; p.panel0 = p0; }   // network input read in place
      p.bias = reinterpret_cast<const float*>(c->arena + s.offBias);
      p.ctrd = reinterpret_cast<const float*>(c->arena + s.offCtrd);
      p.ctrd8 = (s.prog8Bytes || s.prog8SBytes || s.progH8Bytes) ? reinterpret_cast<const float*>(c->arena + s.offCtrd8) : nullptr;
      p.rows = reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt);
      p.prog = s.progBytes ? reinterpret_cast<const uint16_t*>(c->arena + s.offProg) : nullptr;
      p.H = a.h; p.W = a.w; p.Cin = a.c; p.Ho = b.h; p.Wo = b.w; p.Ct = b.c;
      p.knl = d.knlSiz; p.stride = d.stride; p.pad = d.padSiz; p.grp = d.grpCnt;
      p.M = s.M; p.Cs = s.Cs; p.K = s.K; p.pd = s.P; p.relu = fuseRelu ? 1 : 0; p.panels = panels;
      // few-image kernel unless the layer's shape is outside what it covers (a tap window x K that does not fit its LDS
      // table): the panel kernel handles every shape set_layer_shape accepts
      p.splitFrom = 0; p.splitZ = 1; p.partial = nullptr;
      p.nSeg = 0; p.progS = s.progSBytes ? reinterpret_cast<const uint16_t*>(c->arena + s.offProgS) : nullptr;
      s.lastFrom = -1; s.lastZ = 1;
      if (s.P > 1) {                       // more than 128 code words per sub-space: pseudo sub-spaces, exact-builder kernel in every mode
        e = qk_conv_aprx(p, 0, st);
        break;
      }
      e = small ? qk_conv_small(p, live, st) : hipErrorInvalidValue;
      // fp16 table storage (QCNN_OPT_LUT_MODE = 2): the eight-wave tile kernel in its fp16 form wherever the layer's shape has one
      // (K = 128, complete 4- / 8-dim sub-spaces, > 64 channels per group); QCNN_OPT_SYM8 = 0 keeps every layer in the 16-wave
      // kernels, which round the same entries and keep them in f32 slots (same sums, same bits: the tests compare the two)
      // QCNN_OPT_LUT_MODE = 3 keeps the running sums as packed fp16 as well (twice the tile per wave); layers without an fp16
      // form round their entries and keep fp32 sums in both modes
      if (e == hipErrorInvalidValue && c->lutMode >= 2 && c->sym8 && s.prog8Bytes && !inNchw) {
        if (ensure_f16_program(c, l, st)) return 1;
        p.progS = c->lutMode == 3 ? s.prog8A : s.prog8H;
        s.lastFrom = c->lutMode == 3 ? -8 : -7; s.lastZ = 1;   // reported by qcnn_get_layer_split as (-7 / -8 fp16 sums, 1)
        e = qk_conv_sym8(p, st, c->lutMode == 3 ? 2 : 1);
        break;
      }
      if (e == hipErrorInvalidValue) {
        // Which kernel family runs this launch, and how it is cut: the planner (qcnn_planner.h) prices every eligible family for
        // this launch geometry — cached per layer —, its decision rules pick one.  MFMA builders only: the exact builder keeps
        // the tile kernel and the reference's summation order.
        if (c->lutMode >= 1 && (c->split || c->slide || c->sym || c->sym8 || c->half8)) {
          const size_t share = kConvPartialFloats / (size_t)nsub;
          QkPlanOptions o = {};
          o.split = c->split; o.slide = c->slide; o.sym = c->sym; o.sym8 = c->sym8; o.half8 = c->half8;
          o.lutMode = c->lutMode; o.inNchw = inNchw ? 1 : 0; o.scratchFloats = share;
          o.concurrent = (nsub > 1 && panelsAll > panels) ? 1 : 0;
          o.hasSlide16 = s.progSBytes != 0; o.hasSym16 = s.progYBytes != 0; o.hasSym8 = s.prog8Bytes != 0;
          o.hasSym8Slide = s.prog8SBytes != 0; o.hasHalf8 = s.progH8Bytes != 0; o.hasHalf8Slide = s.progH8SBytes != 0;
          const long long key = (((((((((long long)(nsub > 1 ? panelsAll : 0) * 4096 + panels) * 8 + nsub) * 2 + (c->split ? 1 : 0)) * 4 + c->slide) * 4 + c->sym) * 4 +
                                  c->lutMode) * 2 + (inNchw ? 1 : 0)) * 8 + c->sym8) * 4 + c->half8;
          auto it = s.plans.find(key);
          if (it == s.plans.end()) {
            // Sub-batches on several streams run CONCURRENTLY: the tail of one sub-batch's launch fills with the other's workgroups
            // (that is what the streams are for), so the family is chosen for the panels of the whole forward — planned per
            // sub-batch, a 1000-image forward on two streams took the kernels of a 500-image one (conv3 / conv4 back on the 16-wave
            // tile kernel) and lost what the overlap gained: 103.8 k images/s against 107 k with the one-stream plan's kernels.
            ConvParams pp = p;
            if (o.concurrent) pp.panels = panelsAll;
            it = s.plans.emplace(key, qk_plan_conv(pp, o)).first;
          }
          const QkConvChoice ch = qk_choose_conv(it->second, o);
          // partial sums of split tiles: scratch allocated by the first split launch of this context; when the device has no
          // memory left for it (large maps at a large batch) the tiles run whole, which needs none — never a failed forward
          auto partial = [&]() -> float* {
            if (!c->convPartial && !c->noConvPartial && hipMalloc(&c->convPartial, kConvPartialFloats * sizeof(float)) != hipSuccess) {
              (void)hipGetLastError();
              c->convPartial = nullptr; c->noConvPartial = true;
            }
            return c->convPartial ? c->convPartial + share * sub : nullptr;
          };
          auto segments = [&]() {
            p.nSeg = ch.nSeg; s.segN = ch.nSeg;
            for (int i = 0; i <= ch.nSeg; ++i) { p.segBeg[i] = ch.segBeg[i]; s.segBeg[i] = ch.segBeg[i]; }
          };
          s.lastFrom = ch.family; s.lastZ = 1;         // what qcnn_get_layer_split reports: (family code, slices / segments)
          bool launched = true;
          switch (ch.family) {
            case QK_FAM_HALF8_SLIDE:
              p.progS = reinterpret_cast<const uint16_t*>(c->arena + s.offProgH8S);
              segments(); s.lastZ = ch.nSeg;
              e = qk_conv_half8_slide(p, st);
              break;
            case QK_FAM_HALF8:
              p.progS = reinterpret_cast<const uint16_t*>(c->arena + s.offProgH8);
              e = qk_conv_half8(p, st);
              break;
            case QK_FAM_SYM8_SLIDE:
              p.progS = reinterpret_cast<const uint16_t*>(c->arena + s.offProg8S);
              segments(); s.lastZ = ch.nSeg;
              e = qk_conv_sym8_slide(p, st);
              break;
            case QK_FAM_SYM8:
              p.progS = reinterpret_cast<const uint16_t*>(c->arena + s.offProg8);
              if (ch.Z > 1) {
                if (float* ps = partial()) { p.splitFrom = 0; p.splitZ = ch.Z; p.partial = ps; s.lastZ = ch.Z; }
              }
              e = qk_conv_sym8(p, st);
              break;
            case QK_FAM_SYM16:
              p.progS = reinterpret_cast<const uint16_t*>(c->arena + s.offProgY);
              e = qk_conv_sym(p, st);
              break;
            case QK_FAM_SLIDE16:
              segments(); s.lastZ = ch.nSeg;
              launched = false;                          // k_conv_aprx<.., SLIDE> below (p.progS = the sliding program)
              break;
            default:                                     // tile kernel, whole or with a split tail
              s.lastFrom = -1;
              if (ch.Z > 1) {
                if (float* ps = partial()) {
                  p.splitFrom = ch.splitFrom; p.splitZ = ch.Z; p.partial = ps;
                  s.lastFrom = ch.splitFrom; s.lastZ = ch.Z;
                }
              }
              launched = false;
              break;
          }
          if (launched) break;
        }
        e = qk_conv_aprx(p, c->lutMode, st);
      }
      break;
    }
    case QCNN_FCNT: {
      if (!s.loaded) return fail(c, "layer %d: parameters not uploaded", l);
      if (s.dense) {                       // precise path (CalcFeatMap_FCntPrec, src/CaffeEva.cc:932-966): a 1x1 conv on a 1x1 map
        DenseParams q;
        q.src = src; q.dst = dst;
        if (s.hasDmap && !flatFcInput) {
          float* flat = c->fcFlat + (size_t)p0 * fm_elems(c, l) * QCNN_PANEL;
          e = qk_permute_rows(src, flat, reinterpret_cast<const int*>(c->arena + s.offDmap), a.h * a.w * a.c, panels, live, st);
          if (e != hipSuccess) break;
          q.src = flat;
        }
        q.bias = reinterpret_cast<const float*>(c->arena + s.offBias);
        q.wt = reinterpret_cast<const float*>(c->arena + s.offDense);
        q.H = 1; q.W = 1; q.Cin = a.h * a.w * a.c; q.Ho = 1; q.Wo = 1; q.Ct = b.c;
        q.knl = 1; q.stride = 1; q.pad = 0; q.grp = 1;
        q.relu = fuseRelu ? 1 : 0; q.panels = panels;
        e = qk_dense(q, st);
        break;
      }
      if (decoded_fc(c, l)) {              // one-dim sub-spaces: decoded code words on the matrix pipe (qcnn_decoded.hip)
        FcDecParams q;
        q.src = src; q.dst = dst; q.partial = nullptr;
        q.bias = reinterpret_cast<const float*>(c->arena + s.offBias);
        q.wdec = reinterpret_cast<const float*>(c->arena + s.offDec);
        q.D = a.h * a.w * a.c; q.Ct = b.c; q.S = s.decS;
        q.relu = fuseRelu ? 1 : 0; q.panels = panels; q.halves = 2;
        // k slices over workgroups change the summation order with the panel count of the launch: QCNN_OPT_SPLIT only (off =
        // batch-size-invariant bits, as for the split conv tiles and the per-launch FC split below)
        int z = c->split ? std::min(qk_fc_dec_slices(q.D, q.Ct, panels, live), kMaxFcSplit) : 1;
        const size_t need = (size_t)z * panels * q.Ct * QCNN_PANEL;
        const size_t poff = (size_t)kMaxFcSplit * p0 * c->fcMaxCt * QCNN_PANEL;    // every sub-batch has its own slab
        if (z > 1 && poff + need <= c->fcPartialElems) q.partial = c->fcPartial + poff; else z = 1;
        s.lastFrom = -3; s.lastZ = z;        // reported by qcnn_get_layer_split as (-3, k slices over workgroups)
        e = qk_fc_dec(q, z, live, st);
        if (e == hipSuccess && z > 1)
          e = qk_sum_partials(q.partial, dst, z, (size_t)panels * q.Ct * QCNN_PANEL, q.relu, st);
        break;
      }
      s.lastFrom = -1; s.lastZ = 1;
      FcParams p;
      p.src = src; p.dst = dst;
      p.bias = reinterpret_cast<const float*>(c->arena + s.offBias);
      p.ctrd = reinterpret_cast<const float*>(c->arena + s.offCtrd);
      p.rows = reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt);
      p.cbn = (s.cbnBytes && c->packedFc) ? reinterpret_cast<const uint8_t*>(c->arena + s.offCbn) : nullptr;
      p.cbnBits = s.cbnBits;
      if (s.hasDmap && !flatFcInput) {   // NHWC -> consumption order (NCHW flatten) into the scratch map
        float* flat = c->fcFlat + (size_t)p0 * fm_elems(c, l) * QCNN_PANEL;
        e = qk_permute_rows(src, flat, reinterpret_cast<const int*>(c->arena + s.offDmap), a.h * a.w * a.c,
                            panels, live, st);
        if (e != hipSuccess) break;
        p.src = flat;
      }
      p.D = a.h * a.w * a.c; p.Ct = b.c; p.M = s.M; p.Cs = s.Cs; p.K = s.K; p.pd = s.P;
      p.relu = fuseRelu ? 1 : 0; p.panels = panels;
      // Split the sub-space axis over workgroups when the (channel chunk x panel) grid cannot fill the
      // chip; the exact builder keeps one pass so that the summation order stays the reference's.
      p.msplit = 1; p.partial = nullptr;
      if (s.P > 1) {                       // pseudo sub-spaces: one pass of the exact-builder kernel
        e = qk_fc_aprx(p, 0, st);
        break;
      }
      if (small && s.K % 4 == 0 && (size_t)live * s.M * s.K <= c->fcPartialElems) {
        p.partial = c->fcPartial;          // few images: the tables are materialised in the partial-sum scratch
        e = qk_fc_small(p, live, st);      // (K not a multiple of 4: the panel kernel below)
        if (e != hipErrorInvalidValue) break;
        p.partial = nullptr;
      }
      // k_fc_sym8: 768 channels per workgroup.  A launch of one or two panels stays with the 12-wave kernel's 384 (measured,
      // AlexNet fc6 / fc7 per 125 images: 0.092 / 0.053 against 0.108 / 0.070 ms; 250: 0.148 / 0.076 against 0.150 / 0.082; 500:
      // 0.304 / 0.135 against 0.259 / 0.129) — under QCNN_OPT_SPLIT only, whose results may depend on the batch size
      const bool fc8h = s.progF8Bytes && c->sym8 && c->lutMode >= 2 && !small;      // fp16 table storage (3: fp16 sums too): always the eight-wave form
      const bool fc8 = fc8h || (s.progF8Bytes && c->sym8 && c->lutMode == 1 && !small && (c->sym8 >= 2 || !c->split || panels >= 3));
      if (c->lutMode >= 1) {
        const int G = qcnn_stage_group(s.K);
        const int stages = (s.M + G - 1) / G;
        // batch-independent choice (a given image must produce the same bits in any batch): the split count
        // that fills 256 CUs best at the design point of 8 panels (1000 images) while every workgroup keeps
        // >= 24 stages (>= 12 when that leaves a single panel — one GPU's share of a sharded batch — on fewer than 64
        // CUs); ties go to fewer splits.  (A grid of chunks x splits x panels workgroups runs in
        // ceil(grid / 256) rounds: 528 workgroups cost as much as 768.)
        const int cpb = qk_fc_channels_per_block(p.Ct);
        const int chunks = fc8 ? qk_fc_sym8_chunks(p.Ct) : (p.Ct + cpb - 1) / cpb;
        auto pick = [&](int minStages) {
          int best = 1;
          double bestFill = 0.0;
          for (int cand = 1; cand <= kMaxFcSplit; ++cand) {
            if (cand > 1 && stages / cand < minStages) break;
            const int grid = chunks * cand * 8;
            const double fill = (double)grid / (256.0 * ((grid + 255) / 256));
            if (fill > bestFill + 1e-9) { bestFill = fill; best = cand; }
          }
          return best;
        };
        int ms = pick(24);
        if (chunks * ms < 64) ms = pick(12);     // few channel chunks (a 1000-way classifier): a single panel would sit on < 64 CUs
        if (c->split && chunks * ms * panels < 2 * 256) {
          // QCNN_OPT_SPLIT: a launch of a few panels (one GPU's share of a sharded batch) picks the split for ITS panel
          // count (the bits of an image then depend on the batch size, to rounding): >= 8 stages per workgroup, fewest
          // rounds of 256 workgroups x stages each, ties to fewer splits
          int best = ms;
          double bestT = 1e30;
          for (int cand = 1; cand <= kMaxFcSplit; ++cand) {
            if (cand > 1 && stages / cand < 8) break;
            const int grid = chunks * cand * panels;
            const double t = (double)((grid + 255) / 256) * ((double)((stages + cand - 1) / cand) + 10.0) + 0.5 * cand;
            if (t < bestT - 1e-9) { bestT = t; best = cand; }
          }
          ms = best;
        }
        const size_t need = (size_t)ms * panels * p.Ct * QCNN_PANEL;
        const size_t poff = (size_t)kMaxFcSplit * p0 * c->fcMaxCt * QCNN_PANEL;    // every sub-batch has its own slab
        if (ms > 1 && poff + need <= c->fcPartialElems) { p.msplit = ms; p.partial = c->fcPartial + poff; }
      }
      if (fc8) {
        // every workgroup along the sub-space axis needs a stage: the count the launcher will accept for this split
        const int stagesF = s.M / 4, per = (stagesF + p.msplit - 1) / p.msplit;
        p.msplit = (stagesF + per - 1) / per;
        if (p.msplit == 1) p.partial = nullptr;
        s.lastFrom = fc8h ? (c->lutMode == 3 ? -8 : -7) : -5; s.lastZ = p.msplit;   // reported by qcnn_get_layer_split as (-5 / -7 fp16 tables / -8 fp16 sums, splits of the sub-space axis)
        if (fc8h && ensure_f16_program(c, l, st)) return 1;
        e = qk_fc_sym8(p, fc8h ? s.progF8H : reinterpret_cast<const uint16_t*>(c->arena + s.offProgF8),
                       reinterpret_cast<const float*>(c->arena + s.offCtrdF), st, fc8h ? (c->lutMode == 3 ? 2 : 1) : 0);
      } else {
        e = qk_fc_aprx(p, c->lutMode, st);
      }
      if (e == hipSuccess && p.msplit > 1)
        e = qk_sum_partials(p.partial, dst, p.msplit, (size_t)panels * p.Ct * QCNN_PANEL, p.relu, st);
      break;
    }
    case QCNN_POOL:
      e = qk_pool(src, dst, panels, a.h, a.w, a.c, b.h, b.w, d.knlSiz, d.stride, d.padSiz, live, st);
      break;
    case QCNN_RELU:
      e = qk_relu(src, dst, (size_t)panels * fm_elems(c, l) * QCNN_PANEL, st);
      break;
    case QCNN_LORN:
      e = qk_lrn(src, dst, panels, a.h * a.w, a.c, d.lrnSiz, d.lrnAlp, d.lrnBet, d.lrnIni, live, st);
      break;
    case QCNN_DRPT:   // test-time dropout is a copy (src/CaffeEva.cc:1091-1096); only reached by qcnn_run_layer
      e = hipMemcpyAsync(dst, src, (size_t)panels * fm_elems(c, l) * QCNN_PANEL * sizeof(float),
                         hipMemcpyDeviceToDevice, st);
      break;
    case QCNN_SMAX:
      e = qk_softmax(src, dst, panels, a.h * a.w * a.c, live, st);
      break;
    default:
      return fail(c, "layer %d: invalid layer type %d", l, d.type);
  }
  if (e != hipSuccess) return fail(c, "layer %d (type %d) launch failed: %s", l, d.type, hipGetErrorString(e));
  return 0;
}

// accumulate the recorded event pairs into the per-layer sums (blocks until the recorded work is done)
int drain_profile(QcnnCtx* c) {
  if (c->profPending.empty()) { c->profForwards += c->profCount; c->profCount = 0; return 0; }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (const QcnnCtx::ProfRec& r : c->profPending) {
    float ms = 0.0f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[r.slot], c->ev[r.slot + 1]));
    c->profSum[r.layer] += ms;
    c->profLaunches[r.layer] += 1;
  }
  c->profForwards += c->profCount;
  c->profCount = 0;
  c->profPending.clear();
  return 0;
}

// The layers of one forward.  The batch is cut into up to nStreams sub-batches of whole panels; sub-batch 0
// runs on the context's stream, the others on auxiliary streams forked from / joined to it with events, so
// that the LDS-bound conv/FC kernels of one sub-batch overlap the HBM-bound glue kernels of another and the
// last dispatch round of one kernel is filled by the next.  Every image still sees exactly the same
// arithmetic (panels are independent), so results do not depend on the number of streams.
// Can the first layer's builders read the NCHW network input in place (no pack kernel, no packed copy of the input)?
// Fast path only (layer-for-layer mode keeps fm[0] for dumps); a conv layer with <= 4 input channels per group (one
// sub-space of <= 4 dims: exactly what the operand loads of one stage touch) and K = 128 or the exact builder.
// workgroups a fused LRN + pool launch must have
// (192: one panel of AlexNet's LRN1 + pool1 — 196 workgroups — fuses: 0.091 against 0.104 ms; LRN2 + pool2 at one / two panels —
// 64 / 128 workgroups — must not: 0.21 against 0.065 ms)
int lrn_pool_min_blocks() {
#ifdef QCNN_EXPERIMENT     // variant builds only (scripts/build_variant.sh -DQCNN_EXPERIMENT)
  static const int v = [] { const char* e = getenv("QCNN_LRNPOOL_MIN"); return (e && atoi(e) > 0) ? atoi(e) : 192; }();
  return v;
#else
  return 192;
#endif
}

bool direct_input(const QcnnCtx* c, int n) {
  if (c->keepAll || c->L == 0 || c->layers[0].type != QCNN_CONV) return false;
  // the images of THIS forward inside 4 GiB: the in-place kernels keep image offsets in 32 bits
  if ((unsigned long long)n * c->inC * c->inH * c->inW * sizeof(float) >= (1ull << 32)) return false;
  // a first layer that runs through its decoded code words reads packed panels — unless its kernel has the NCHW form
  // (k_conv_dec_nchw); batches of one to three images go to the few-image table kernel below, which reads NCHW densely
  if (decoded_layer(c, 0) && !(c->smallBatch && c->lutMode == 1 && n <= kSmallBatchMax)) return c->directDec && c->shapes[0].decNV > 0;
  const QcnnLayerDesc& d = c->layers[0];
  // table kernels: ONE sub-space (a second one would be fetched from channel planes past the group's own, for the last
  // image past the caller's buffer)
  if (c->shapes[0].dense || c->shapes[0].M != 1) return false;
  return c->inC / d.grpCnt <= 4 && (c->lutMode == 0 || c->shapes[0].K == 128);
}

// pa / pb: the panels [pa, pb) of the batch this call runs (pb < 0: all of them) — a large host batch goes through in
// chunks whose uploads overlap the previous chunk's layers, all chunks writing into the same whole-batch feature maps
int run_layers(QcnnCtx* c, int n, const float* inNchw = nullptr, int pa = 0, int pb = -1) {
  const int panelsAll = (n + QCNN_PANEL - 1) / QCNN_PANEL;
  if (pb < 0) pb = panelsAll;
  const int panels = pb - pa;
  const int ns = std::max(1, std::min(std::min(c->nStreams, kMaxStreams), panels));
  // A batch of a few images: conv/FC by the channel-lane kernels, glue kernels on the live lanes only.  Only in the
  // default f32 mode: the exact builder's point is the reference's summation order (which only the panel kernels
  // keep), and modes 2 / 3 study properties of the panel kernels' table builders.
  const bool small = c->smallBatch && c->lutMode == 1 && n <= kSmallBatchMax;
  const int live = panelsAll == 1 ? n : QCNN_PANEL;
  if (c->profile && c->profCount == kProfRing && drain_profile(c)) return 1;   // ring full: fold it into the sums
  const bool prof = c->profile != 0;
  // pointer table of THIS call (aliases; nullptr = not materialised by this call: the network input read in place, the
  // normalised map of a fused LRN + pool pair); it becomes / is merged into the context's table at the end
  std::vector<float*> fm(c->L + 1, nullptr);
  fm[0] = inNchw ? nullptr : c->fmBuf[0];
  // LRN + the 3x3 / stride 2 / pad 0 max-pool behind it run as one kernel on the fast path when every sub-batch
  // fills the chip with it; the normalised map is then not materialised (qcnn_get_feature_map reports it missing)
  auto lrn_pool = [&](int l) {
    if (c->keepAll || small || l + 1 >= c->L) return false;
    const QcnnLayerDesc& a = c->layers[l];
    const QcnnLayerDesc& b = c->layers[l + 1];
    if (a.type != QCNN_LORN || b.type != QCNN_POOL || (a.lrnSiz != 5 && a.lrnSiz != 3)) return false;
    if (b.knlSiz != 3 || b.stride != 2 || b.padSiz != 0) return false;
    return (long long)qk_lrn_pool_blocks(c->dims[l + 2].h, c->dims[l + 2].w) * (panels / ns) >= lrn_pool_min_blocks();
  };
  for (int l = 0; l < c->L; ++l) {              // pointer table (aliases) — identical for every sub-batch
    const int type = c->layers[l].type;
    const bool prevFused = l > 0 && !c->keepAll && type == QCNN_RELU &&
                           (c->layers[l - 1].type == QCNN_CONV || c->layers[l - 1].type == QCNN_FCNT);
    fm[l + 1] = (type == QCNN_DRPT || prevFused) ? fm[l] : c->fmBuf[l + 1];
    if (lrn_pool(l)) fm[l + 1] = nullptr;
  }
  for (int k = 0; k < ns - 1; ++k) {                 // auxiliary streams of the sub-batches, created when first needed
    if (!c->aux[k]) HIP_TRY(c, hipStreamCreateWithFlags(&c->aux[k], hipStreamNonBlocking));
    if (!c->evJoin[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->evJoin[k], hipEventDisableTiming));
  }
  if (ns > 1) {
    HIP_TRY(c, hipEventRecord(c->evFork, c->stream));
    for (int k = 1; k < ns; ++k) HIP_TRY(c, hipStreamWaitEvent(c->aux[k - 1], c->evFork, 0));
  }
  for (int l = 0; l < c->L; ++l) {              // layer-major issue order: the streams advance together
    const int type = c->layers[l].type;
    if (fm[l + 1] == fm[l] && fm[l] != nullptr) continue;   // alias: copy semantics, no traffic
    if (fm[l] == nullptr && l > 0) continue;                               // the pool of a fused LRN + pool pair
    const bool lrnPool = fm[l + 1] == nullptr;
    const bool fuse = !c->keepAll && (type == QCNN_CONV || type == QCNN_FCNT) && l + 1 < c->L &&
                      c->layers[l + 1].type == QCNN_RELU;
    for (int k = 0; k < ns; ++k) {
      const int p0 = pa + (int)((long long)panels * k / ns), p1 = pa + (int)((long long)panels * (k + 1) / ns);
      if (p1 <= p0) continue;
      hipStream_t st = k == 0 ? c->stream : c->aux[k - 1];
      const bool direct = l == 0 && inNchw != nullptr;
      const float* src = direct ? nullptr : fm[l] + (size_t)p0 * fm_elems(c, l) * QCNN_PANEL;
      float* dst = lrnPool ? fm[l + 2] + (size_t)p0 * fm_elems(c, l + 2) * QCNN_PANEL
                           : fm[l + 1] + (size_t)p0 * fm_elems(c, l + 1) * QCNN_PANEL;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (prof) {
        const size_t slot = (((size_t)c->profCount * kMaxStreams + k) * c->L + l) * 2;
        e0 = c->ev[slot]; e1 = c->ev[slot + 1];
        HIP_TRY(c, hipEventRecord(e0, st));
      }
      if (lrnPool) {
        const QcnnLayerDesc& d = c->layers[l];
        const hipError_t e = qk_lrn_pool(src, dst, p1 - p0, c->dims[l].h, c->dims[l].w, c->dims[l].c, c->dims[l + 2].h,
                                         c->dims[l + 2].w, d.lrnSiz, d.lrnAlp, d.lrnBet, d.lrnIni, live, st);
        if (e != hipSuccess) return fail(c, "layer %d (LRN + pool): %s", l, hipGetErrorString(e));
      } else if (launch_layer(c, l, src, dst, p1 - p0, fuse, false, p0, st, direct ? inNchw : nullptr, n, live, small, k, ns, panels)) {
        return 1;
      }
      if (prof) {
        HIP_TRY(c, hipEventRecord(e1, st));
        c->profPending.push_back(QcnnCtx::ProfRec{(((size_t)c->profCount * kMaxStreams + k) * c->L + l) * 2, l});
      }
    }
  }
  for (int k = 1; k < ns; ++k) {
    HIP_TRY(c, hipEventRecord(c->evJoin[k - 1], c->aux[k - 1]));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evJoin[k - 1], 0));
  }
  if (prof) c->profCount++;
  // what can be read back afterwards: a map exists only if every chunk of the batch materialised it
  if (pa == 0 || (int)c->lastFm.size() != c->L + 1) {
    c->lastFm = fm;
  } else {
    for (int l = 0; l <= c->L; ++l)
      if (fm[l] == nullptr) c->lastFm[l] = nullptr;
  }
  c->lastN = n;
  return 0;
}

}  // namespace

extern "C" {

int qcnn_abi_version(void) { return QCNN_ABI_VERSION; }

int qcnn_device_count(int* count) {
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) { *count = 0; return fail(nullptr, "hipGetDeviceCount -> %s", hipGetErrorString(e)); }
  return 0;
}

const char* qcnn_last_error(const QcnnCtx* ctx) { return ctx ? ctx->err.c_str() : g_createError.c_str(); }

int qcnn_ctx_create(int device_id, void* stream, QcnnCtx** out) {
  if (!out) return fail(nullptr, "qcnn_ctx_create: out == NULL");
  *out = nullptr;
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0)
    return fail(nullptr, "no HIP device available (%s); this library has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device_id < 0 || device_id >= cnt) return fail(nullptr, "device %d out of range [0, %d)", device_id, cnt);
  e = hipSetDevice(device_id);
  if (e != hipSuccess) return fail(nullptr, "hipSetDevice(%d) -> %s", device_id, hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device_id);
  if (e != hipSuccess) return fail(nullptr, "hipGetDeviceProperties -> %s", hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
  QcnnCtx* c = new QcnnCtx;
  c->device = device_id;
  if (stream) {
    c->stream = static_cast<hipStream_t>(stream);
  } else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(nullptr, "hipStreamCreate -> %s", hipGetErrorString(e)); }
    c->ownStream = true;
  }
  // The copy stream is created HERE, second: the runtime spreads a process's streams over a handful of hardware queues
  // (four by default) in creation order, and a copy stream that shares its queue with the compute stream serialises
  // "upload batch b + 1" in front of "layers of batch b" (measured: 25 ms instead of 15 ms per 1000-image batch).  The
  // auxiliary compute streams are created on demand (run_layers), so the default two-stream set-up uses three queues.
  e = hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    if (c->ownStream) (void)hipStreamDestroy(c->stream);
    delete c;
    return fail(nullptr, "hipStreamCreate (copy stream) -> %s", hipGetErrorString(e));
  }
  *out = c;
  return 0;
}

int qcnn_ctx_destroy(QcnnCtx* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  free_model(c);
  for (int k = 0; k < kMaxStreams - 1; ++k) {
    if (c->aux[k]) (void)hipStreamDestroy(c->aux[k]);
    if (c->evJoin[k]) (void)hipEventDestroy(c->evJoin[k]);
  }
  if (c->evFork) (void)hipEventDestroy(c->evFork);
  for (int k = 0; k < 2; ++k) {
    if (c->evCopied[k]) (void)hipEventDestroy(c->evCopied[k]);
    if (c->evFreed[k]) (void)hipEventDestroy(c->evFreed[k]);
    if (c->evDone[k]) (void)hipEventDestroy(c->evDone[k]);
  }
  for (hipEvent_t e : c->evChunk) (void)hipEventDestroy(e);
  if (c->copyStream) (void)hipStreamDestroy(c->copyStream);
  if (c->ownStream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

int qcnn_set_option(QcnnCtx* c, int option, int value) {
  switch (option) {
    case QCNN_OPT_LUT_MODE: if (value < 0 || value > 3) return fail(c, "LUT mode must be 0 (exact), 1 (f32 MFMA), 2 (fp16 table storage) or 3 (fp16 tables and fp16 sums)"); c->lutMode = value; return 0;   // (part of the plan key)
    case QCNN_OPT_KEEP_ALL: c->keepAll = value ? 1 : 0; return 0;
    case QCNN_OPT_PROFILE: c->profile = value ? 1 : 0; return 0;
    case QCNN_OPT_SMALL_BATCH: c->smallBatch = value ? 1 : 0; return 0;
    case QCNN_OPT_SPLIT: c->split = value ? 1 : 0; return 0;
    case QCNN_OPT_DECODE: c->decode = value ? 1 : 0; return 0;
    case QCNN_OPT_HALF8: if (value < 0 || value > 3) return fail(c, "QCNN_OPT_HALF8 must be 0 (off), 1 (planner), 2 (forced tile form) or 3 (forced sliding form)"); c->half8 = value; return 0;
    case QCNN_OPT_SYM8: if (value < 0 || value > 3) return fail(c, "QCNN_OPT_SYM8 must be 0 (off), 1 (planner), 2 (forced tile form) or 3 (forced sliding form)"); c->sym8 = value; return 0;
    case QCNN_OPT_PACKED_FC: c->packedFc = value ? 1 : 0; return 0;
    case QCNN_OPT_DIRECT_DEC: c->directDec = value ? 1 : 0; return 0;
    case QCNN_OPT_SYM: if (value < 0 || value > 2) return fail(c, "QCNN_OPT_SYM must be 0 (off), 1 (planner) or 2 (forced)"); c->sym = value; return 0;
    case QCNN_OPT_SLIDE: if (value < 0 || value > 2) return fail(c, "QCNN_OPT_SLIDE must be 0 (off), 1 (planner) or 2 (forced)"); c->slide = value; return 0;
    case QCNN_OPT_HOST_CHUNK:
      if (value < 0) return fail(c, "host chunk must be >= 0 panels");
      c->hostChunk = value; return 0;
    case QCNN_OPT_STREAMS:
      if (value < 1 || value > kMaxStreams) return fail(c, "streams must be in [1, %d]", kMaxStreams);
      c->nStreams = value; return 0;
    default: return fail(c, "unknown option %d", option);
  }
}

int qcnn_ctx_device(const QcnnCtx* c) { return c->device; }
void* qcnn_ctx_stream(const QcnnCtx* c) { return c->stream; }

int qcnn_sync(QcnnCtx* c) {
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

int qcnn_model_begin(QcnnCtx* c, int layer_cnt, const QcnnLayerDesc* layers, int in_c, int in_h, int in_w) {
  HIP_TRY(c, hipSetDevice(c->device));
  free_model(c);
  if (layer_cnt <= 0 || !layers) return fail(c, "qcnn_model_begin: empty layer table");
  c->L = layer_cnt; c->inC = in_c; c->inH = in_h; c->inW = in_w;
  c->layers.assign(layers, layers + layer_cnt);
  c->shapes.assign(layer_cnt, LayerShape());
  c->dims.assign(layer_cnt + 1, FmDims{0, 0, 0});
  c->firstFc = -1;
  int h = in_h, w = in_w, ch = in_c;
  c->dims[0] = FmDims{h, w, ch};
  for (int l = 0; l < layer_cnt; ++l) {           // feature-map size rule, src/CaffeEva.cc:357-391
    const QcnnLayerDesc& d = layers[l];
    switch (d.type) {
      case QCNN_CONV:
        if (d.grpCnt <= 0 || d.stride <= 0 || d.knlSiz <= 0 || ch % d.grpCnt || d.knlCnt % d.grpCnt)
          return fail(c, "layer %d: bad conv geometry", l);
        h = conv_out(h, d.knlSiz, d.stride, d.padSiz); w = conv_out(w, d.knlSiz, d.stride, d.padSiz); ch = d.knlCnt;
        break;
      case QCNN_POOL:
        if (d.stride <= 0 || d.knlSiz <= 0) return fail(c, "layer %d: bad pool geometry", l);
        h = pool_out(h, d.knlSiz, d.stride, d.padSiz); w = pool_out(w, d.knlSiz, d.stride, d.padSiz);
        break;
      case QCNN_FCNT:
        if (c->firstFc < 0) c->firstFc = l;
        h = 1; w = 1; ch = d.nodCnt;
        break;
      case QCNN_RELU: case QCNN_LORN: case QCNN_DRPT: case QCNN_SMAX: break;
      default: return fail(c, "layer %d: invalid layer type %d", l, d.type);
    }
    if (h <= 0 || w <= 0 || ch <= 0) return fail(c, "layer %d: empty feature map", l);
    c->dims[l + 1] = FmDims{h, w, ch};
  }
  return 0;
}

int qcnn_model_set_layer_shape(QcnnCtx* c, int layer, int M, int K, int Cs) {
  if (layer < 0 || layer >= c->L) return fail(c, "layer %d out of range", layer);
  const QcnnLayerDesc& d = c->layers[layer];
  if (d.type != QCNN_CONV && d.type != QCNN_FCNT) return fail(c, "layer %d carries no parameters", layer);
  if (M <= 0 || K <= 0 || K > QCNN_MAX_K_FILE || Cs <= 0 || Cs > QCNN_MAX_CS)
    return fail(c, "layer %d: unsupported quantisation shape M=%d K=%d Cs=%d (K <= %d, Cs <= %d)", layer, M, K, Cs,
                QCNN_MAX_K_FILE, QCNN_MAX_CS);
  const int D = (d.type == QCNN_CONV) ? c->dims[layer].c / d.grpCnt : (int)fm_elems(c, layer);
  if ((size_t)M * Cs < (size_t)D) return fail(c, "layer %d: M*Cs = %d does not cover %d input dims", layer, M * Cs, D);
  if ((M - 1) * Cs >= D) return fail(c, "layer %d: sub-space %d starts beyond the %d input dims", layer, M - 1, D);
  const int Ct = c->dims[layer + 1].c;
  const int Ctg = (d.type == QCNN_CONV) ? Ct / d.grpCnt : Ct;
  if (Ctg % 2) return fail(c, "layer %d: %d output channels per group is not even", layer, Ctg);
  LayerShape& s = c->shapes[layer];
  s.Mfile = M; s.Kfile = K; s.Cs = Cs;
  // more code words than a 128-row LDS stage holds (the reference's uint8 assignments allow 256, include/FileIO.h:128-166): P pseudo
  // sub-spaces of <= 127 code words + one all-zero row each over the SAME dims; an assignment names its code word in one of them and
  // the zero row in the others — the same sums (x + 0 = x), through the exact-builder kernels (ConvParams::pd)
  s.P = K > QCNN_MAX_K ? (K + 126) / 127 : 1;
  s.M = M * s.P;
  s.K = s.P > 1 ? QCNN_MAX_K : K;
  return 0;
}

int qcnn_model_set_layer_dense(QcnnCtx* c, int layer) {
  if (layer < 0 || layer >= c->L) return fail(c, "layer %d out of range", layer);
  const QcnnLayerDesc& d = c->layers[layer];
  if (d.type != QCNN_CONV && d.type != QCNN_FCNT) return fail(c, "layer %d carries no parameters", layer);
  if (c->committed) return fail(c, "qcnn_model_set_layer_dense must precede qcnn_model_commit");
  c->shapes[layer] = LayerShape();
  c->shapes[layer].dense = true;
  return 0;
}

int qcnn_model_set_layer_weights(QcnnCtx* c, int layer, const float* bias, const float* weights_file) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->committed) return fail(c, "qcnn_model_commit must precede qcnn_model_set_layer_weights");
  if (layer < 0 || layer >= c->L) return fail(c, "layer %d out of range", layer);
  const QcnnLayerDesc& d = c->layers[layer];
  LayerShape& s = c->shapes[layer];
  if (!s.dense) return fail(c, "layer %d was not declared dense (qcnn_model_set_layer_dense)", layer);
  const int Ct = c->dims[layer + 1].c;
  const int grp = (d.type == QCNN_CONV) ? d.grpCnt : 1;
  const int taps = (d.type == QCNN_CONV) ? d.knlSiz * d.knlSiz : 1;
  const int Cg = (d.type == QCNN_CONV) ? c->dims[layer].c / grp : (int)fm_elems(c, layer);
  const int Ctg = Ct / grp;
  // file layout [Ct][Cg][kh][kw] (convKnl) / [Ct][D] (fcntWei)  ->  [grp][tap][Cg][Ctg], output channel innermost
  std::vector<float> wt(s.denseFloats);
  for (int g = 0; g < grp; ++g)
    for (int ch = 0; ch < Ctg; ++ch)
      for (int ci = 0; ci < Cg; ++ci)
        for (int t = 0; t < taps; ++t)
          wt[(((size_t)g * taps + t) * Cg + ci) * Ctg + ch] = weights_file[(((size_t)(g * Ctg + ch)) * Cg + ci) * taps + t];
  HIP_TRY(c, hipMemcpyAsync(c->arena + s.offBias, bias, sizeof(float) * Ct, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->arena + s.offDense, wt.data(), sizeof(float) * wt.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  s.loaded = true;
  return 0;
}

int qcnn_model_arena_bytes(QcnnCtx* c, size_t* bytes) {
  if (plan_arena(c)) return 1;
  *bytes = c->arenaBytes;
  return 0;
}

int qcnn_model_arena_ptr(QcnnCtx* c, void** dev_ptr, size_t* bytes) {
  if (!c->committed) return fail(c, "model not committed");
  if (dev_ptr) *dev_ptr = c->arena;
  if (bytes) *bytes = c->arenaBytes;
  return 0;
}

namespace {
// two 64-bit sums over the arena's 32-bit words: the plain sum and a position-weighted one (a permutation of blocks changes it)
__global__ __launch_bounds__(256) void k_arena_checksum(const uint32_t* __restrict__ w, size_t n, unsigned long long* __restrict__ out) {
  unsigned long long a = 0, b = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned long long v = w[i];
    a += v;
    b += v * (unsigned long long)(i % 65521u + 1u);
  }
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], a); atomicAdd(&out[1], b); }
}
}  // namespace

/* Checksum of the packed parameter arena as it lies on the device (after uploads / a broadcast): sum2[0] = sum of its 32-bit
 * words, sum2[1] = position-weighted sum.  Ranks whose arenas hold the same bytes report the same pair — what a sharded run
 * compares before it trusts a broadcast.  Blocking. */
int qcnn_model_arena_checksum(QcnnCtx* c, unsigned long long* sum2) {
  if (!c->committed) return fail(c, "model not committed");
  if (!sum2) return fail(c, "qcnn_model_arena_checksum: sum2 == NULL");
  HIP_TRY(c, hipSetDevice(c->device));
  unsigned long long* d = nullptr;
  HIP_TRY(c, hipMalloc(&d, 2 * sizeof(unsigned long long)));
  hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_arena_checksum, dim3(1024), dim3(256), 0, c->stream, reinterpret_cast<const uint32_t*>(c->arena),
                       c->arenaBytes / 4, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(sum2, d, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(c, "arena checksum -> %s", hipGetErrorString(e));
  return 0;
}

int qcnn_model_commit(QcnnCtx* c, int max_batch, void* dev_arena) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->committed) return fail(c, "model already committed");
  if (max_batch <= 0) return fail(c, "max_batch must be positive");
  if (plan_arena(c)) return 1;
  c->maxBatch = max_batch;
  c->maxPanels = (max_batch + QCNN_PANEL - 1) / QCNN_PANEL;
  if (dev_arena) {
    c->arena = static_cast<char*>(dev_arena);
    c->ownArena = false;
  } else {
    HIP_TRY(c, hipMalloc(&c->arena, c->arenaBytes));   // arenaBytes already includes the slack
    c->ownArena = true;
    HIP_TRY(c, hipMemsetAsync(c->arena, 0, c->arenaBytes, c->stream));
  }
  c->fmBuf.assign(c->L + 1, nullptr);
  for (int l = 0; l <= c->L; ++l) {
    if (l > 0 && c->layers[l - 1].type == QCNN_DRPT) continue;   // always an alias of its input
    const size_t bytes = (size_t)c->maxPanels * fm_elems(c, l) * QCNN_PANEL * sizeof(float) + kSlack;
    HIP_TRY(c, hipMalloc(&c->fmBuf[l], bytes));
  }
  {
    size_t maxCt = 0;
    for (int l = 0; l < c->L; ++l)
      if (c->layers[l].type == QCNN_FCNT) maxCt = std::max<size_t>(maxCt, c->dims[l + 1].c);
    c->fcMaxCt = maxCt;
    c->fcPartialElems = (size_t)kMaxFcSplit * c->maxPanels * maxCt * QCNN_PANEL;
    if (c->fcPartialElems) HIP_TRY(c, hipMalloc(&c->fcPartial, c->fcPartialElems * sizeof(float)));
    if (c->firstFc >= 0 && c->shapes[c->firstFc].hasDmap)
      HIP_TRY(c, hipMalloc(&c->fcFlat, (size_t)c->maxPanels * fm_elems(c, c->firstFc) * QCNN_PANEL * sizeof(float) + kSlack));
  }
  c->ev.resize((size_t)kProfRing * kMaxStreams * c->L * 2);
  for (hipEvent_t& e : c->ev) HIP_TRY(c, hipEventCreate(&e));
  c->profSum.assign(c->L, 0.0);
  c->profLaunches.assign(c->L, 0);
  c->profCount = 0; c->profForwards = 0; c->profPending.clear();
  if (!c->evFork) HIP_TRY(c, hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));   // aux streams: on demand (run_layers)
  // first-FC flatten map: consumption index d = (ch*H + y)*W + x  ->  NHWC row (y*W + x)*C + ch  (src/CaffeEva.cc:187-189)
  for (int l = 0; l < c->L; ++l) {
    const LayerShape& s = c->shapes[l];
    if (!s.hasDmap) continue;
    const FmDims& a = c->dims[l];
    std::vector<int> map(fm_elems(c, l));
    for (int ch = 0; ch < a.c; ++ch)
      for (int y = 0; y < a.h; ++y)
        for (int x = 0; x < a.w; ++x) map[((size_t)ch * a.h + y) * a.w + x] = (y * a.w + x) * a.c + ch;
    HIP_TRY(c, hipMemcpyAsync(c->arena + s.offDmap, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  c->committed = true;
  return 0;
}

namespace {
// bias + code book (PrepCtrdBuf permutation; the eight-wave kernels' operand orders) into the arena
int upload_bias_ctrd(QcnnCtx* c, int layer, const float* bias, const float* ctrd_file) {
  LayerShape& s = c->shapes[layer];
  const int Ct = c->dims[layer + 1].c;
  const int M = s.M, K = s.K, Cs = s.Cs;
  // PrepCtrdBuf: [M][K][Cs] -> [M][Cs][K]  (src/CaffeEva.cc:556-557)
  std::vector<float> ctrd((size_t)M * Cs * K);
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k)
      for (int dd = 0; dd < Cs; ++dd) ctrd[((size_t)m * Cs + dd) * K + k] = ctrd_file[((size_t)m * K + k) * Cs + dd];
  std::vector<float> ctrdF;
  if (s.progF8Bytes) {                // the eight-wave FC kernel's operand order (K = 32, Cs = 4)
    ctrdF.resize((size_t)M * Cs * K);
    for (int m = 0; m < M; ++m)
      for (int dd = 0; dd < Cs; ++dd)
        for (int k = 0; k < K; ++k) ctrdF[qk_ctrdf_index(m, dd, k)] = ctrd[((size_t)m * Cs + dd) * K + k];
    HIP_TRY(c, hipMemcpyAsync(c->arena + s.offCtrdF, ctrdF.data(), ctrdF.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  }
  std::vector<float> ctrd8;
  if (s.prog8Bytes || s.prog8SBytes || s.progH8Bytes) {  // the eight-wave symmetric kernel's operand order (K = 128, Cs = 4 or 8)
    ctrd8.resize((size_t)M * Cs * K);
    for (int m = 0; m < M; ++m)
      for (int dd = 0; dd < Cs; ++dd)
        for (int k = 0; k < K; ++k) ctrd8[qk_ctrd8_index(m, dd, k, Cs / 4)] = ctrd[((size_t)m * Cs + dd) * K + k];
    HIP_TRY(c, hipMemcpyAsync(c->arena + s.offCtrd8, ctrd8.data(), ctrd8.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  }
  HIP_TRY(c, hipMemcpyAsync(c->arena + s.offBias, bias, sizeof(float) * Ct, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->arena + s.offCtrd, ctrd.data(), sizeof(float) * ctrd.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));      // the host vectors die with this scope
  return 0;
}
}  // namespace

namespace {
// FC layer: the assignments (file order [Ct][M], 0-based code words) bit-packed exactly as a .cbn payload of `bits` bits per
// element (include/FileIO.h:299-341: 4096-byte blocks of floor(32768 / bits) values, MSB first, no value across a block)
// into the arena: the resident form the few-image kernel reads in place
int upload_packed_assignments(QcnnCtx* c, int layer, const uint8_t* asmt_file) {
  const LayerShape& s = c->shapes[layer];
  if (!s.cbnBytes) return 0;
  const size_t n = (size_t)c->dims[layer + 1].c * s.M;
  const int bits = s.cbnBits;
  const size_t per = 4096 * 8 / (size_t)bits;
  std::vector<uint8_t> blocks(s.cbnBytes, 0);
  for (size_t e = 0; e < n; ++e) {
    const size_t bit0 = (e % per) * bits;
    uint8_t* b = blocks.data() + (e / per) * 4096 + (bit0 >> 3);
    const unsigned w = (unsigned)asmt_file[e] << (16 - (bit0 & 7) - bits);      // <= 8 bits: at most two bytes
    b[0] |= (uint8_t)(w >> 8);
    if (w & 0xffu) b[1] |= (uint8_t)(w & 0xffu);
  }
  HIP_TRY(c, hipMemcpyAsync(c->arena + s.offCbn, blocks.data(), blocks.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

// rows table of a conv layer (already in the arena, same stream) -> program table
hipError_t build_program(QcnnCtx* c, int layer, const QkSlots& sl) {
  const QcnnLayerDesc& d = c->layers[layer];
  const LayerShape& s = c->shapes[layer];
  hipError_t e = hipSuccess;
  if (s.decKp < 0)                  // FC layer with one-dim sub-spaces
    e = qk_decode_fc_weights(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), reinterpret_cast<const float*>(c->arena + s.offCtrd),
                             reinterpret_cast<float*>(c->arena + s.offDec), sl, (int)fm_elems(c, layer), s.K,
                             c->dims[layer + 1].c, s.decS, c->stream);
  if (s.decKp > 0)                  // one sub-space of <= 4 dims: the code word every assignment names (qcnn_decoded.hip)
    e = qk_decode_weights(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), reinterpret_cast<const float*>(c->arena + s.offCtrd),
                          reinterpret_cast<float*>(c->arena + s.offDec), sl, d.knlSiz, c->dims[layer].c, s.K,
                          c->dims[layer + 1].c, s.decKp, s.decS, c->stream);
  if (e == hipSuccess && s.decKp > 0 && s.decNV)
    e = qk_decode_weights_nchw(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), reinterpret_cast<const float*>(c->arena + s.offCtrd),
                               reinterpret_cast<float*>(c->arena + s.offDecN), sl, d.knlSiz, c->dims[layer].c, s.K,
                               c->dims[layer + 1].c, s.decNV, c->dims[layer + 1].c, c->stream);
  if (e == hipSuccess && s.progF8Bytes)        // eight-wave FC kernel: uint16 offsets in its channel order
    e = qk_build_program_fc8(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), reinterpret_cast<uint16_t*>(c->arena + s.offProgF8), sl,
                             c->dims[layer + 1].c, s.M, c->stream);
  if (e == hipSuccess && s.prog8Bytes) {       // eight-wave symmetric kernel: its own layout of the same table
    const int Ct = c->dims[layer + 1].c;
    e = qk_build_program8(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), reinterpret_cast<uint16_t*>(c->arena + s.offProg8), sl,
                          qk_conv_sym8_config(c->dims[layer].c, d.grpCnt, Ct, s.M, s.Cs, s.K), Ct / d.grpCnt, d.grpCnt, d.knlSiz,
                          d.stride, s.M, c->stream);
  }
  if (e == hipSuccess && s.progH8Bytes) {      // half-panel eight-wave kernel
    const int Ct = c->dims[layer + 1].c;
    e = qk_build_program_h8(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), reinterpret_cast<uint16_t*>(c->arena + s.offProgH8), sl,
                            qk_conv_half8_config(c->dims[layer].c, d.grpCnt, Ct, s.M, s.Cs, s.K), Ct / d.grpCnt, d.grpCnt, d.knlSiz,
                            d.stride, s.M, c->stream);
  }
  if (e == hipSuccess && s.progH8SBytes) {     // ... and its sliding form
    const int Ct = c->dims[layer + 1].c;
    e = qk_build_program_h8(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), reinterpret_cast<uint16_t*>(c->arena + s.offProgH8S), sl,
                            qk_conv_half8_slide_config(c->dims[layer].c, d.grpCnt, Ct, s.M, s.Cs, s.K, d.knlSiz, d.stride), Ct / d.grpCnt,
                            d.grpCnt, d.knlSiz, d.stride, s.M, c->stream);
  }
  if (e == hipSuccess && s.prog8SBytes) {      // ... and the program of its sliding form
    const int Ct = c->dims[layer + 1].c;
    e = qk_build_program8(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), reinterpret_cast<uint16_t*>(c->arena + s.offProg8S), sl,
                          qk_conv_sym8_slide_config(c->dims[layer].c, d.grpCnt, Ct, s.M, s.Cs, s.K, d.knlSiz, d.stride), Ct / d.grpCnt,
                          d.grpCnt, d.knlSiz, d.stride, s.M, c->stream);
  }
  if (e == hipSuccess && s.progYBytes) {       // symmetric kernel: the (8 channels per wave, 2x2 tile) layout of the same table
    const QkSlots s8 = qk_make_slots(sl.C, sl.groups, 8);
    e = qk_build_program(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt), reinterpret_cast<uint16_t*>(c->arena + s.offProgY),
                         sl, s8, qk_conv_program(s8, d.knlSiz, d.stride), d.knlSiz, d.stride, s.M, c->stream);
  }
  if (e != hipSuccess || !s.progBytes) return e;
  e = qk_build_program(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt),
                                  reinterpret_cast<uint16_t*>(c->arena + s.offProg), sl, sl,
                                  qk_conv_program(sl, d.knlSiz, d.stride), d.knlSiz, d.stride, s.M, c->stream);
  if (e == hipSuccess && s.progSBytes) {
    const QkSlide sc = qk_slide_config(sl.C, sl.groups, d.knlSiz, d.stride);
    e = qk_build_program(reinterpret_cast<const uint8_t*>(c->arena + s.offAsmt),
                         reinterpret_cast<uint16_t*>(c->arena + s.offProgS), sl, sc.sl,
                         qk_conv_program_slide(sc.sl, sc.ns, sc.nc, d.knlSiz, d.stride), d.knlSiz, d.stride, s.M, c->stream, 1);
  }
  return e;
}
}  // namespace

int qcnn_model_set_layer_params(QcnnCtx* c, int layer, const float* bias, const float* ctrd_file,
                                const uint8_t* asmt_file) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->committed) return fail(c, "qcnn_model_commit must precede qcnn_model_set_layer_params");
  if (layer < 0 || layer >= c->L) return fail(c, "layer %d out of range", layer);
  const QcnnLayerDesc& d = c->layers[layer];
  LayerShape& s = c->shapes[layer];
  if (s.K <= 0) return fail(c, "layer %d carries no quantised parameters", layer);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drop_f16_programs(s);
  const int Ct = c->dims[layer + 1].c;
  std::vector<float> ctrdX;
  std::vector<uint8_t> asmtX;
  if (s.P > 1) {
    // file layout (Mfile, Kfile) -> pseudo sub-spaces (M = P * Mfile, K = 128): pseudo sub-space j of m holds code words 127 j .. 127 j +
    // 126 in its rows 0 .. 126 and zeros in row 127; an assignment v becomes row v % 127 of pseudo sub-space v / 127, row 127 elsewhere
    const size_t tapsX = (d.type == QCNN_CONV) ? (size_t)d.knlSiz * d.knlSiz : 1;
    const int P = s.P, Mf = s.Mfile, Kf = s.Kfile, Cs = s.Cs;
    ctrdX.assign((size_t)s.M * 128 * Cs, 0.0f);
    for (int m = 0; m < Mf; ++m)
      for (int k = 0; k < Kf; ++k)
        for (int dd = 0; dd < Cs; ++dd)
          ctrdX[(((size_t)m * P + k / 127) * 128 + k % 127) * Cs + dd] = ctrd_file[((size_t)m * Kf + k) * Cs + dd];
    asmtX.resize((size_t)Ct * tapsX * s.M);
    for (size_t e = 0; e < (size_t)Ct * tapsX * Mf; ++e) {
      const unsigned v = asmt_file[e];
      if ((int)v >= Kf) return fail(c, "layer %d: assignment %u >= K = %d", layer, v, Kf);
      for (int j = 0; j < P; ++j) asmtX[e * P + j] = (uint8_t)((int)(v / 127) == j ? v % 127 : 127);
    }
    ctrd_file = ctrdX.data();
    asmt_file = asmtX.data();
  }
  const int M = s.M, K = s.K;
  // PrepAsmtBuf: conv [Ct][kh][kw][M] -> [kh][kw][M][Ct] (:585-586); FC [Ct][M] -> [M][Ct] (:610-611).  Stored as
  // the one-byte SLOT of the code word's row inside a LUT stage (row = (m % G) * K + index < 128; LDS offset = slot * 64),
  // with the channel axis in the order the gather waves consume it (QkSlots); padding entries point at slot 0.
  const int G = qcnn_stage_group(K);
  const size_t taps = (d.type == QCNN_CONV) ? (size_t)d.knlSiz * d.knlSiz : 1;
  const int groups = (d.type == QCNN_CONV) ? d.grpCnt : 1;
  const QkSlots sl = (d.type == QCNN_CONV) ? qk_conv_slots(Ct / groups, groups) : qk_fc_slots(Ct);
  std::vector<uint8_t> asmt(s.asmtBytes + QCNN_ROWS_PAD, 0);
  for (int ch = 0; ch < Ct; ++ch) {
    const int entry = qk_slot_entry(sl, ch / sl.C, ch % sl.C);
    for (size_t t = 0; t < taps; ++t)
      for (int m = 0; m < M; ++m) {
        const uint8_t v = asmt_file[((size_t)ch * taps + t) * M + m];
        if (v >= K) return fail(c, "layer %d: assignment %u >= K = %d", layer, (unsigned)v, K);
        asmt[(t * M + m) * sl.rowStride + entry] = (uint8_t)qcnn_row_slot((m % G) * K + v);
      }
  }
  if (upload_bias_ctrd(c, layer, bias, ctrd_file)) return 1;
  if (upload_packed_assignments(c, layer, asmt_file)) return 1;
  HIP_TRY(c, hipMemcpyAsync(c->arena + s.offAsmt, asmt.data(), asmt.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, build_program(c, layer, sl));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  s.loaded = true;
  return 0;
}

int qcnn_model_set_layer_params_cbn(QcnnCtx* c, int layer, const float* bias, const float* ctrd_file,
                                    const uint8_t* cbn_blocks, size_t cbn_bytes, int bits) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->committed) return fail(c, "qcnn_model_commit must precede qcnn_model_set_layer_params_cbn");
  if (layer < 0 || layer >= c->L) return fail(c, "layer %d out of range", layer);
  const QcnnLayerDesc& d = c->layers[layer];
  LayerShape& s = c->shapes[layer];
  if (s.K <= 0) return fail(c, "layer %d carries no parameters", layer);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  drop_f16_programs(s);
  if (bits < 1 || bits > 8) return fail(c, "layer %d: %d bits per assignment (1..8 supported)", layer, bits);
  const int Ct = c->dims[layer + 1].c;
  const size_t taps = (d.type == QCNN_CONV) ? (size_t)d.knlSiz * d.knlSiz : 1;
  if (s.P > 1) {                        // more than 128 code words: unpacked here, expanded into pseudo sub-spaces by the byte path
    const size_t nF = (size_t)Ct * taps * s.Mfile, perF = 4096 * 8 / (size_t)bits;
    if (cbn_bytes < (nF + perF - 1) / perF * 4096) return fail(c, "layer %d: %zu bytes of packed assignments, %zu needed", layer, cbn_bytes, (nF + perF - 1) / perF * 4096);
    std::vector<uint8_t> vals(nF);
    for (size_t e = 0; e < nF; ++e) {
      const size_t bit0 = (e % perF) * bits;
      const uint8_t* b = cbn_blocks + (e / perF) * 4096 + (bit0 >> 3);
      const unsigned w = ((unsigned)b[0] << 8) | (unsigned)b[(bit0 & 7) + bits > 8 ? 1 : 0];
      vals[e] = (uint8_t)((w >> (16 - (bit0 & 7) - bits)) & ((1u << bits) - 1u));
    }
    return qcnn_model_set_layer_params(c, layer, bias, ctrd_file, vals.data());
  }
  const int groups = (d.type == QCNN_CONV) ? d.grpCnt : 1;
  const QkSlots sl = (d.type == QCNN_CONV) ? qk_conv_slots(Ct / groups, groups) : qk_fc_slots(Ct);
  const size_t n = (size_t)Ct * taps * s.M;
  const size_t per = 4096 * 8 / (size_t)bits;
  const size_t need = (n + per - 1) / per * 4096;
  if (cbn_bytes < need) return fail(c, "layer %d: %zu bytes of packed assignments, %zu needed", layer, cbn_bytes, need);
  if (upload_bias_ctrd(c, layer, bias, ctrd_file)) return 1;
  if (s.cbnBytes && bits != s.cbnBits) {           // a stream of another width: re-packed at the layer's own width for the resident copy
    std::vector<uint8_t> vals(n);
    bool bad = false;
    for (size_t e = 0; e < n; ++e) {
      const size_t bit0 = (e % per) * bits;
      const uint8_t* b = cbn_blocks + (e / per) * 4096 + (bit0 >> 3);
      const unsigned w = ((unsigned)b[0] << 8) | (unsigned)b[(bit0 & 7) + bits > 8 ? 1 : 0];
      vals[e] = (uint8_t)((w >> (16 - (bit0 & 7) - bits)) & ((1u << bits) - 1u));
      bad = bad || vals[e] >= s.K;
    }
    if (bad) return fail(c, "layer %d: an assignment >= K = %d in the packed stream", layer, s.K);
    if (upload_packed_assignments(c, layer, vals.data())) return 1;
  }
  uint8_t* dev = nullptr;
  int* bad = nullptr;
  HIP_TRY(c, hipMalloc(&dev, need + sizeof(int)));
  bad = reinterpret_cast<int*>(dev + need);
  hipError_t e = hipMemcpyAsync(dev, cbn_blocks, need, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && s.cbnBytes && bits == s.cbnBits)       // the payload itself is the resident packed form
    e = hipMemcpyAsync(c->arena + s.offCbn, dev, std::min(need, s.cbnBytes), hipMemcpyDeviceToDevice, c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(bad, 0, sizeof(int), c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(c->arena + s.offAsmt, 0, s.asmtBytes + QCNN_ROWS_PAD, c->stream);   // padding entries -> row 0
  if (e == hipSuccess)
    e = qk_decode_cbn(dev, bits, n, Ct, (int)taps, s.M, s.K, sl, reinterpret_cast<uint8_t*>(c->arena + s.offAsmt), bad, c->stream);
  if (e == hipSuccess) e = build_program(c, layer, sl);
  int flag = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&flag, bad, sizeof(int), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(c, "layer %d: device-side assignment decode failed: %s", layer, hipGetErrorString(e));
  if (flag) return fail(c, "layer %d: an assignment >= K = %d in the packed stream", layer, s.K);
  s.loaded = true;
  return 0;
}

/* Mark every conv/FC layer as loaded without uploading: the arena was filled by a broadcast. */
int qcnn_model_mark_loaded(QcnnCtx* c) {
  if (!c->committed) return fail(c, "model not committed");
  // The arena may have been REfilled (a re-upload on rank 0 + a second broadcast, or a caller-owned arena written again): the
  // lazily built fp16 program tables (QCNN_OPT_LUT_MODE = 2 / 3) were derived from the OLD assignment bytes — drop them, they are
  // rebuilt from the arena on the next forward that needs them.  Nothing may still be reading them.
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < kMaxStreams - 1; ++k)
    if (c->aux[k]) HIP_TRY(c, hipStreamSynchronize(c->aux[k]));
  for (int l = 0; l < c->L; ++l) {
    drop_f16_programs(c->shapes[l]);
    if (c->shapes[l].K > 0 || c->shapes[l].dense) c->shapes[l].loaded = true;
  }
  return 0;
}

int qcnn_fm_dims(QcnnCtx* c, int l, int* hwc3) {
  if (l < 0 || l > c->L) return fail(c, "feature map %d out of range", l);
  hwc3[0] = c->dims[l].h; hwc3[1] = c->dims[l].w; hwc3[2] = c->dims[l].c;
  return 0;
}

namespace {
// layers + output conversion of a forward whose input panel (fmBuf[0]) has just been enqueued
// output conversion of a finished layer loop: probabilities [n][classes], top-5 [n][5]
int forward_outputs(QcnnCtx* c, int n, float* prob_dev, uint16_t* top5_dev) {
  const int classes = (int)fm_elems(c, c->L);
  hipError_t e;
  if (prob_dev) {
    e = qk_unpack_rows(c->lastFm[c->L], prob_dev, n, classes, c->stream);
    if (e != hipSuccess) return fail(c, "output unpack launch failed: %s", hipGetErrorString(e));
  }
  if (top5_dev) {
    e = qk_top5(c->lastFm[c->L], top5_dev, n, classes, c->stream);
    if (e != hipSuccess) return fail(c, "top-5 launch failed: %s", hipGetErrorString(e));
  }
  return 0;
}
int forward_tail(QcnnCtx* c, int n, float* prob_dev, uint16_t* top5_dev, const float* inNchw = nullptr) {
  if (run_layers(c, n, inNchw)) return 1;
  return forward_outputs(c, n, prob_dev, top5_dev);
}
}  // namespace

int qcnn_forward(QcnnCtx* c, const float* in_nchw_dev, int n, float* prob_dev, uint16_t* top5_dev) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->committed) return fail(c, "model not committed");
  if (n <= 0 || n > c->maxBatch) return fail(c, "batch %d outside (0, %d]", n, c->maxBatch);
  if (direct_input(c, n)) return forward_tail(c, n, prob_dev, top5_dev, in_nchw_dev);   // conv1's builders read it in place
  hipError_t e = qk_pack_nchw(in_nchw_dev, c->fmBuf[0], n, c->inC, c->inH, c->inW, c->stream);
  if (e != hipSuccess) return fail(c, "input pack launch failed: %s", hipGetErrorString(e));
  return forward_tail(c, n, prob_dev, top5_dev);
}

int qcnn_forward_u8(QcnnCtx* c, const uint8_t* in_u8_dev, int src_h, int src_w, const float* mean_dev, int n,
                    float* prob_dev, uint16_t* top5_dev) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->committed) return fail(c, "model not committed");
  if (n <= 0 || n > c->maxBatch) return fail(c, "batch %d outside (0, %d]", n, c->maxBatch);
  if (src_h < c->inH || src_w < c->inW)
    return fail(c, "source images %dx%d are smaller than the network input %dx%d", src_h, src_w, c->inH, c->inW);
  hipError_t e = qk_pack_u8(in_u8_dev, mean_dev, c->fmBuf[0], n, c->inC, c->inH, c->inW, src_h, src_w, c->stream);
  if (e != hipSuccess) return fail(c, "input pack launch failed: %s", hipGetErrorString(e));
  return forward_tail(c, n, prob_dev, top5_dev);
}

int qcnn_host_register(void* ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(nullptr, "qcnn_host_register: empty range");
  const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, "hipHostRegister(%zu bytes) -> %s", bytes, hipGetErrorString(e)); }
  return 0;
}

int qcnn_host_alloc(size_t bytes, void** out) {
  if (!out || !bytes) return fail(nullptr, "qcnn_host_alloc: empty request");
  *out = nullptr;
  const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
  if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return fail(nullptr, "hipHostMalloc(%zu bytes) -> %s", bytes, hipGetErrorString(e)); }
  return 0;
}

int qcnn_host_free(void* ptr) {
  if (!ptr) return 0;
  const hipError_t e = hipHostFree(ptr);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, "hipHostFree -> %s", hipGetErrorString(e)); }
  return 0;
}

int qcnn_host_unregister(void* ptr) {
  if (!ptr) return 0;
  const hipError_t e = hipHostUnregister(ptr);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, "hipHostUnregister -> %s", hipGetErrorString(e)); }
  return 0;
}

// The reference's image loop (src/CaffeEva.cc:151-211) classifies one batch after the other; here the upload of batch
// b + 1 (copy stream, second input buffer) runs under the layers of batch b, and the results come back through pinned
// buffers one batch late, so that neither direction of PCIe is ever waited for by the kernels.
int qcnn_forward_host_batches(QcnnCtx* c, const float* const* in_host, const int* n, int nb, float* const* prob_host,
                              uint16_t* const* top5_host) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->committed) return fail(c, "model not committed");
  if (nb <= 0 || !in_host || !n) return fail(c, "qcnn_forward_host_batches: no batches");
  for (int b = 0; b < nb; ++b)
    if (n[b] <= 0 || n[b] > c->maxBatch || !in_host[b]) return fail(c, "batch %d: %d images outside (0, %d]", b, n[b], c->maxBatch);
  if (ensure_pipeline(c)) return 1;
  const size_t inE = fm_elems(c, 0);
  const size_t classes = fm_elems(c, c->L);
  float* const dIn[2] = {c->stageIn, c->stageIn1};
  // the copy stream may not touch an input buffer before everything already enqueued on the compute stream has read it
  for (int k = 0; k < 2; ++k) { HIP_TRY(c, hipEventRecord(c->evFreed[k], c->stream)); c->freedValid[k] = true; }
  auto upload = [&](int b) -> int {
    const int k = b & 1;
    if (c->freedValid[k]) HIP_TRY(c, hipStreamWaitEvent(c->copyStream, c->evFreed[k], 0));
    HIP_TRY(c, hipMemcpyAsync(dIn[k], in_host[b], inE * n[b] * sizeof(float), hipMemcpyHostToDevice, c->copyStream));
    HIP_TRY(c, hipEventRecord(c->evCopied[k], c->copyStream));
    return 0;
  };
  auto collect = [&](int b) -> int {
    const int k = b & 1;
    HIP_TRY(c, hipEventSynchronize(c->evDone[k]));
    if (prob_host && prob_host[b]) memcpy(prob_host[b], c->pinProb[k], (size_t)n[b] * classes * sizeof(float));
    if (top5_host && top5_host[b]) memcpy(top5_host[b], c->pinTop5[k], (size_t)n[b] * 5 * sizeof(uint16_t));
    return 0;
  };
  // QCNN_DEBUG_PIPELINE=1: time every upload and every batch's kernels with events and print the schedule afterwards
  static const bool dbg = [] { const char* e = getenv("QCNN_DEBUG_PIPELINE"); return e && atoi(e) != 0; }();
  std::vector<hipEvent_t> dbgEv;
  if (dbg) {
    dbgEv.resize((size_t)nb * 4 + 1);
    for (hipEvent_t& e : dbgEv) HIP_TRY(c, hipEventCreate(&e));
    HIP_TRY(c, hipEventRecord(dbgEv[(size_t)nb * 4], c->stream));
  }
  auto uploadT = [&](int b) -> int {
    if (!dbg) return upload(b);
    const int k = b & 1;
    if (c->freedValid[k]) HIP_TRY(c, hipStreamWaitEvent(c->copyStream, c->evFreed[k], 0));
    HIP_TRY(c, hipEventRecord(dbgEv[(size_t)b * 4], c->copyStream));
    HIP_TRY(c, hipMemcpyAsync(dIn[k], in_host[b], inE * n[b] * sizeof(float), hipMemcpyHostToDevice, c->copyStream));
    HIP_TRY(c, hipEventRecord(dbgEv[(size_t)b * 4 + 1], c->copyStream));
    HIP_TRY(c, hipEventRecord(c->evCopied[k], c->copyStream));
    return 0;
  };
  if (uploadT(0)) return 1;
  for (int b = 0; b < nb; ++b) {
    const int k = b & 1;
    if (b + 1 < nb && uploadT(b + 1)) return 1;
    const bool wantProb = prob_host && prob_host[b], wantTop5 = top5_host && top5_host[b];
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evCopied[k], 0));
    if (dbg) HIP_TRY(c, hipEventRecord(dbgEv[(size_t)b * 4 + 2], c->stream));
    if (qcnn_forward(c, dIn[k], n[b], wantProb ? c->stageOut : nullptr, wantTop5 ? c->stageTop5 : nullptr)) return 1;
    if (dbg) HIP_TRY(c, hipEventRecord(dbgEv[(size_t)b * 4 + 3], c->stream));
    HIP_TRY(c, hipEventRecord(c->evFreed[k], c->stream));
    if (wantProb)
      HIP_TRY(c, hipMemcpyAsync(c->pinProb[k], c->stageOut, (size_t)n[b] * classes * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (wantTop5)
      HIP_TRY(c, hipMemcpyAsync(c->pinTop5[k], c->stageTop5, (size_t)n[b] * 5 * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipEventRecord(c->evDone[k], c->stream));
    if (b >= 1 && collect(b - 1)) return 1;
  }
  if (collect(nb - 1)) return 1;
  HIP_TRY(c, hipStreamSynchronize(c->copyStream));
  if (dbg) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int b = 0; b < nb; ++b) {
      float t[4];
      for (int j = 0; j < 4; ++j) HIP_TRY(c, hipEventElapsedTime(&t[j], dbgEv[(size_t)nb * 4], dbgEv[(size_t)b * 4 + j]));
      fprintf(stderr, "[qcnn pipeline] batch %d (%d images): upload %.2f -> %.2f ms, layers %.2f -> %.2f ms\n", b, n[b], t[0], t[1],
              t[2], t[3]);
    }
    for (hipEvent_t e : dbgEv) (void)hipEventDestroy(e);
  }
  return 0;
}

int qcnn_forward_host(QcnnCtx* c, const float* in_nchw_host, int n, float* prob_host, uint16_t* top5_host) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->committed) return fail(c, "model not committed");
  if (n <= 0 || n > c->maxBatch) return fail(c, "batch %d outside (0, %d]", n, c->maxBatch);
  const size_t inE = fm_elems(c, 0);
  const int classes = (int)fm_elems(c, c->L);
  if (c->hostChunk > 0 && n >= 2 * c->hostChunk * QCNN_PANEL) {
    // A batch of at least two chunks (QCNN_OPT_HOST_CHUNK panels each, default two) goes through chunk by chunk: every chunk is uploaded on the copy stream
    // into its place of the batch's input buffer, and its layers start as soon as it has arrived — the upload of chunk
    // k + 1 (a DMA transfer from pinned / registered memory, a staged copy otherwise) runs under the layers of chunk k.
    // All chunks write into the same whole-batch feature maps, so dumps and results are those of one launch.
    if (ensure_pipeline(c)) return 1;
    const int panelsAll = (n + QCNN_PANEL - 1) / QCNN_PANEL;
    const int chunkPanels = c->hostChunk;
    const int nc = (panelsAll + chunkPanels - 1) / chunkPanels;
    while ((int)c->evChunk.size() < nc) {
      hipEvent_t ev;
      HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      c->evChunk.push_back(ev);
    }
    const bool direct = direct_input(c, n);
    HIP_TRY(c, hipEventRecord(c->evFreed[0], c->stream));            // the buffer may still feed an earlier forward
    HIP_TRY(c, hipStreamWaitEvent(c->copyStream, c->evFreed[0], 0));
    for (int k = 0; k < nc; ++k) {
      const size_t first = (size_t)k * chunkPanels * QCNN_PANEL;
      const size_t cnt = std::min<size_t>((size_t)chunkPanels * QCNN_PANEL, (size_t)n - first);
      HIP_TRY(c, hipMemcpyAsync(c->stageIn + first * inE, in_nchw_host + first * inE, cnt * inE * sizeof(float),
                                hipMemcpyHostToDevice, c->copyStream));
      HIP_TRY(c, hipEventRecord(c->evChunk[k], c->copyStream));
    }
    for (int k = 0; k < nc; ++k) {
      const int pa = k * chunkPanels, pb = std::min(panelsAll, pa + chunkPanels);
      const size_t first = (size_t)pa * QCNN_PANEL;
      const int cnt = (int)std::min<size_t>((size_t)(pb - pa) * QCNN_PANEL, (size_t)n - first);
      HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evChunk[k], 0));
      if (!direct) {
        const hipError_t e = qk_pack_nchw(c->stageIn + first * inE, c->fmBuf[0] + first * inE, cnt, c->inC, c->inH, c->inW, c->stream);
        if (e != hipSuccess) return fail(c, "input pack launch failed: %s", hipGetErrorString(e));
      }
      if (run_layers(c, n, direct ? c->stageIn : nullptr, pa, pb)) return 1;
    }
    if (forward_outputs(c, n, prob_host ? c->stageOut : nullptr, top5_host ? c->stageTop5 : nullptr)) return 1;
    if (prob_host)
      HIP_TRY(c, hipMemcpyAsync(prob_host, c->stageOut, (size_t)n * classes * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (top5_host)
      HIP_TRY(c, hipMemcpyAsync(top5_host, c->stageTop5, (size_t)n * 5 * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
  }
  if (ensure_stage(c)) return 1;
  HIP_TRY(c, hipMemcpyAsync(c->stageIn, in_nchw_host, inE * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  if (qcnn_forward(c, c->stageIn, n, prob_host ? c->stageOut : nullptr, top5_host ? c->stageTop5 : nullptr)) return 1;
  if (prob_host)
    HIP_TRY(c, hipMemcpyAsync(prob_host, c->stageOut, (size_t)n * classes * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (top5_host)
    HIP_TRY(c, hipMemcpyAsync(top5_host, c->stageTop5, (size_t)n * 5 * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

int qcnn_get_layer_output(QcnnCtx* c, int l, int n, float* host_out) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (l < 0 || l > c->L) return fail(c, "feature map %d out of range", l);
  if (n <= 0 || n > c->lastN) return fail(c, "n = %d exceeds the last forward's batch %d", n, c->lastN);
  if ((int)c->lastFm.size() != c->L + 1 || !c->lastFm[l]) return fail(c, "feature map %d is not available", l);
  if (!c->keepAll && l > 0 && (c->layers[l - 1].type == QCNN_CONV || c->layers[l - 1].type == QCNN_FCNT) &&
      l < c->L && c->layers[l].type == QCNN_RELU)
    return fail(c, "feature map %d was fused away (QCNN_OPT_KEEP_ALL = 0)", l);
  if (ensure_stage(c, (size_t)n * fm_elems(c, l))) return 1;
  const int E = (int)fm_elems(c, l);
  hipError_t e = qk_unpack_rows(c->lastFm[l], c->stageOut, n, E, c->stream);
  if (e != hipSuccess) return fail(c, "unpack launch failed: %s", hipGetErrorString(e));
  HIP_TRY(c, hipMemcpyAsync(host_out, c->stageOut, (size_t)n * E * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

int qcnn_get_layer_output_range(QcnnCtx* c, int l, int first, int n, float* host_out) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (l < 0 || l > c->L) return fail(c, "feature map %d out of range", l);
  if (first < 0 || n <= 0 || first + n > c->lastN)
    return fail(c, "images [%d, %d) are not inside the last forward's batch of %d", first, first + n, c->lastN);
  if ((int)c->lastFm.size() != c->L + 1 || !c->lastFm[l]) return fail(c, "feature map %d is not available", l);
  if (!c->keepAll && l > 0 && (c->layers[l - 1].type == QCNN_CONV || c->layers[l - 1].type == QCNN_FCNT) &&
      l < c->L && c->layers[l].type == QCNN_RELU)
    return fail(c, "feature map %d was fused away (QCNN_OPT_KEEP_ALL = 0)", l);
  if (ensure_stage(c, (size_t)(n + QCNN_PANEL) * fm_elems(c, l))) return 1;
  const size_t E = fm_elems(c, l);
  const int p0 = first / QCNN_PANEL;                       // whole panels from the one that holds `first`
  const int cnt = first + n - p0 * QCNN_PANEL;
  hipError_t e = qk_unpack_rows(c->lastFm[l] + (size_t)p0 * E * QCNN_PANEL, c->stageOut, cnt, (int)E, c->stream);
  if (e != hipSuccess) return fail(c, "unpack launch failed: %s", hipGetErrorString(e));
  HIP_TRY(c, hipMemcpyAsync(host_out, c->stageOut + (size_t)(first - p0 * QCNN_PANEL) * E, (size_t)n * E * sizeof(float),
                            hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

int qcnn_run_layer(QcnnCtx* c, int layer, const float* in_host, int n, float* out_host) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->committed) return fail(c, "model not committed");
  if (layer < 0 || layer >= c->L) return fail(c, "layer %d out of range", layer);
  if (n <= 0 || n > c->maxBatch) return fail(c, "batch %d outside (0, %d]", n, c->maxBatch);
  if (ensure_stage(c, (size_t)n * std::max(fm_elems(c, layer), fm_elems(c, layer + 1)))) return 1;
  const int Ein = (int)fm_elems(c, layer), Eout = (int)fm_elems(c, layer + 1);
  const int panels = (n + QCNN_PANEL - 1) / QCNN_PANEL;
  // scratch: reuse the layer's own input/output maps (sized for maxBatch)
  float* src = c->fmBuf[layer] ? c->fmBuf[layer] : c->fmBuf[layer - 1];
  float* dst = c->fmBuf[layer + 1] ? c->fmBuf[layer + 1] : c->stageOut;
  if (c->layers[layer].type == QCNN_DRPT && !c->fmBuf[layer + 1]) {
    // aliased output map: stage through the linear buffer instead
    memcpy(out_host, in_host, (size_t)n * Ein * sizeof(float));
    return 0;
  }
  HIP_TRY(c, hipMemcpyAsync(c->stageIn, in_host, (size_t)n * Ein * sizeof(float), hipMemcpyHostToDevice, c->stream));
  hipError_t e = qk_pack_rows(c->stageIn, src, n, Ein, c->stream);
  if (e != hipSuccess) return fail(c, "pack launch failed: %s", hipGetErrorString(e));
  if (launch_layer(c, layer, src, dst, panels, false, true, 0, c->stream)) return 1;
  e = qk_unpack_rows(dst, c->stageOut, n, Eout, c->stream);
  if (e != hipSuccess) return fail(c, "unpack launch failed: %s", hipGetErrorString(e));
  HIP_TRY(c, hipMemcpyAsync(out_host, c->stageOut, (size_t)n * Eout * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

int qcnn_get_layer_split(QcnnCtx* c, int layer, int* tiles_unsplit, int* slices) {
  if (layer < 0 || layer >= c->L) return fail(c, "layer %d out of range", layer);
  const LayerShape& s = c->shapes[layer];
  if (tiles_unsplit) *tiles_unsplit = s.lastFrom;
  if (slices) *slices = s.lastZ;
  return 0;
}

int qcnn_get_layer_segments(QcnnCtx* c, int layer, int* seg_beg9, int* n_seg) {
  if (layer < 0 || layer >= c->L) return fail(c, "layer %d out of range", layer);
  const LayerShape& s = c->shapes[layer];
  const int n = (s.lastFrom == -2 || s.lastFrom == -6 || s.lastFrom == -10) ? s.segN : 0;
  if (n_seg) *n_seg = n;
  if (seg_beg9)
    for (int i = 0; i < 9; ++i) seg_beg9[i] = (i <= n) ? s.segBeg[i] : 0;
  return 0;
}

int qcnn_get_layer_ms(QcnnCtx* c, float* ms, int* forwards_recorded) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (drain_profile(c)) return 1;
  for (int l = 0; l < c->L; ++l) ms[l] = c->profLaunches[l] ? (float)(c->profSum[l] / c->profLaunches[l]) : 0.0f;
  if (forwards_recorded) *forwards_recorded = c->profForwards;
  return 0;
}

int qcnn_get_layer_total_ms(QcnnCtx* c, double* total_ms, long long* launches, int* forwards_recorded) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (drain_profile(c)) return 1;
  for (int l = 0; l < c->L; ++l) {
    total_ms[l] = c->profSum[l];
    if (launches) launches[l] = c->profLaunches[l];
  }
  if (forwards_recorded) *forwards_recorded = c->profForwards;
  return 0;
}

int qcnn_reset_layer_ms(QcnnCtx* c) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (drain_profile(c)) return 1;
  c->profSum.assign(c->L, 0.0);
  c->profLaunches.assign(c->L, 0);
  c->profForwards = 0;
  return 0;
}

}  // extern "C"
