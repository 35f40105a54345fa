// caffe_eva.cc — host side of CaffeEva: parameter hand-over to the device and the batch loop.
// Mirrors the control flow of the reference's src/CaffeEva.cc (LoadCaffePara :109-149, ExecForwardPass
// :151-261, CalcPredAccu :263-295, DispElpsTime :297-326); the per-layer work is the C-ABI of
// include/qcnn_hip.h.
#include "../../include/CaffeEva.h"

#include <algorithm>

#include "../../include/FileIO.h"
#include "../../include/qcnn_hip.h"

namespace {

const int kLablCntPerData = 5;   // top-5, as src/CaffeEva.cc:25

int envInt(const char* name, int dflt) {
  const char* v = getenv(name);
  if (v == nullptr || *v == '\0') return dflt;
  const int x = atoi(v);
  return x > 0 ? x : dflt;
}

QcnnLayerDesc toDesc(const LayerInfo& li) {
  QcnnLayerDesc d;
  d.type = static_cast<int>(li.type);
  d.padSiz = li.padSiz; d.knlSiz = li.knlSiz; d.knlCnt = li.knlCnt; d.grpCnt = li.grpCnt;
  d.stride = li.stride; d.nodCnt = li.nodCnt; d.lrnSiz = li.lrnSiz;
  d.lrnAlp = li.lrnAlp; d.lrnBet = li.lrnBet; d.lrnIni = li.lrnIni; d.drpRat = li.drpRat;
  return d;
}

}  // namespace

CaffeEva::CaffeEva(void)
    : enblAprx(true), grp_(nullptr), ctx_(nullptr), modelReady_(false), batchSize_(1), batchCnt_(100),
      inflight_(1), imagesDone_(0), keepAll_(false), pinned_(nullptr), pinnedCopy_(nullptr), pinnedImages_(0) {}

CaffeEva::~CaffeEva(void) {
  if (grp_ != nullptr) qcnn_group_destroy(grp_);   // owns the per-device contexts
  unpinDataset();
}

// Where the image block lives while it is classified.  QCNN_PIN_DATASET = "copy" (default): a pinned buffer from the HIP
// runtime (qcnn_host_alloc) that takes over the images from dataLst — uploads from it are DMA transfers at the full
// PCIe rate which run under the previous batch's kernels; "register": dataLst's own storage registered with the runtime
// (no second copy, about half the transfer rate); "0": plain pageable memory (uploads are staged copies).
void CaffeEva::pinDataset(void) {
  unpinDataset();
  if (dataLst.GetEleCnt() <= 0 || dataLst.GetDimCnt() != 4) return;
  const char* env = getenv("QCNN_PIN_DATASET");
  const std::string mode = (env != nullptr && *env != '\0') ? env : "copy";
  // Only the images ExecForwardPass(void) will submit are pinned: QCNN_BATCHES x QCNN_BATCH of them from the start of the
  // dataset (its window rule, src/CaffeEva.cc:170-177: when the batches cover the dataset, all of it) — an ImageNet-sized
  // dataMatTst is neither doubled in host memory nor page-locked for the 100 images that are classified.
  const long long dataCnt = dataLst.GetDimLen(0);
  const long long bs = envInt("QCNN_BATCH", 1), bc = envInt("QCNN_BATCHES", 100);
  const long long batchesInData = bs > 0 ? (dataCnt + bs - 1) / bs : 0;
  const long long used = (bs <= 0 || bc >= batchesInData) ? dataCnt : std::min(dataCnt, bs * bc);
  pinnedImages_ = static_cast<int>(used);
  const size_t bytes = sizeof(float) * static_cast<size_t>(dataLst.GetDimStp(0)) * static_cast<size_t>(used);
  if (mode == "0" || mode == "off") return;
  if (mode == "register") {
    if (qcnn_host_register(dataLst.GetDataPtr(), bytes) == 0) pinned_ = dataLst.GetDataPtr();
    else printf("[INFO] dataset not pinned (%s): uploads are staged copies\n", qcnn_last_error(nullptr));
    return;
  }
  void* buf = nullptr;
  if (qcnn_host_alloc(bytes, &buf) != 0) {
    printf("[INFO] no pinned buffer for the dataset (%s): uploads are staged copies\n", qcnn_last_error(nullptr));
    return;
  }
  memcpy(buf, dataLst.GetDataPtr(), bytes);
  pinnedCopy_ = static_cast<float*>(buf);
}

void CaffeEva::unpinDataset(void) {
  if (pinned_ != nullptr) qcnn_host_unregister(pinned_);
  if (pinnedCopy_ != nullptr) qcnn_host_free(pinnedCopy_);
  pinned_ = nullptr;
  pinnedCopy_ = nullptr;
  pinnedImages_ = 0;
}

bool CaffeEva::fail(const std::string& what) {
  lastError_ = what;
  if (grp_ != nullptr && *qcnn_group_last_error(grp_)) lastError_ += std::string(": ") + qcnn_group_last_error(grp_);
  else if (ctx_ != nullptr && *qcnn_last_error(ctx_)) lastError_ += std::string(": ") + qcnn_last_error(ctx_);
  printf("[ERROR] %s\n", lastError_.c_str());
  return false;
}

void CaffeEva::Init(const bool enblAprxSrc) {
  enblAprx = enblAprxSrc;
  swWall_.Reset();
  if (ctx_ != nullptr && modelReady_) qcnn_reset_layer_ms(ctx_);
}

void CaffeEva::SetModelName(const std::string& modelNameSrc) {
  printf("[CHECK-POINT] entering CaffeEva::SetModelName()\n");
  modelName = modelNameSrc;
}

void CaffeEva::SetModelPath(const std::string& dirPathMainSrc, const std::string& fileNamePfxSrc) {
  printf("[CHECK-POINT] entering CaffeEva::SetModelPath()\n");
  dirPathMain = dirPathMainSrc;
  fileNamePfx = fileNamePfxSrc;
}

bool CaffeEva::LoadDataset(const std::string& dirPathData) {
  printf("[CHECK-POINT] entering CaffeEva::LoadDataset()\n");
  unpinDataset();
  if (!FileIO::ReadBinFile(dirPathData + "/dataMatTst.single.bin", &dataLst)) return false;
  if (!FileIO::ReadBinFile(dirPathData + "/lablVecTst.uint16.bin", &lablVecGrth)) return false;
  if (grp_ != nullptr) pinDataset();     // else: once the device group exists (LoadCaffePara)
  return true;
}

bool CaffeEva::LoadCaffePara(void) {
  printf("[CHECK-POINT] entering CaffeEva::LoadCaffePara()\n");
  modelReady_ = false;

  caffeParaObj.Init(dirPathMain, fileNamePfx);
  if (modelName == "AlexNet") caffeParaObj.ConfigLayer_AlexNet();
  else if (modelName == "CaffeNet") caffeParaObj.ConfigLayer_CaffeNet();
  else if (modelName == "VggCnnS") caffeParaObj.ConfigLayer_VggCnnS();
  else if (modelName == "VGG16") caffeParaObj.ConfigLayer_VGG16();
  else if (modelName == "CaffeNetFGB") caffeParaObj.ConfigLayer_CaffeNetFGB();
  else if (modelName == "CaffeNetFGD") caffeParaObj.ConfigLayer_CaffeNetFGD();
  else {
    printf("[ERROR] unrecognized caffe model name: %s\n", modelName.c_str());
    return false;
  }
  if (!caffeParaObj.LoadLayerPara(enblAprx, ENUM_AsmtEnc::Compact)) return false;
  // A limit of the MI355X build, reported HERE in the reference's convention ([ERROR] + false from LoadCaffePara) instead of
  // surfacing later from the device library: output channels are handled in pairs (the reference itself needs multiples of 8 per
  // group: the unrolled loops of src/CaffeEva.cc:849-858, 1008-1017 over-run otherwise).  (Up to 256 code words per sub-space — all
  // the reference's uint8 assignments can name, include/FileIO.h:128-166 — are supported: above 128 the device library cuts a
  // sub-space into pseudo sub-spaces, qcnn_model_set_layer_shape.)
  if (enblAprx) {
    for (int l = 0; l < caffeParaObj.layerCnt; ++l) {
      const LayerInfo& li = caffeParaObj.layerInfoLst[l];
      if (li.type != ENUM_LyrType::Conv && li.type != ENUM_LyrType::FCnt) continue;
      const int perGrp = (li.type == ENUM_LyrType::Conv) ? li.knlCnt / (li.grpCnt > 0 ? li.grpCnt : 1) : li.nodCnt;
      if (perGrp % 2 != 0) {
        printf("[ERROR] layer #%d: %d output channels per group; this build needs an even count\n", l + 1, perGrp);
        return false;
      }
    }
  }
  batchSize_ = envInt("QCNN_BATCH", 1);
  batchCnt_ = envInt("QCNN_BATCHES", 100);
  // images handed to the devices at once: all of them up to QCNN_MAX_INFLIGHT (QCNN_COALESCE=0: one logical batch at a time)
  const long long all = static_cast<long long>(batchSize_) * batchCnt_;
  const int cap = envInt("QCNN_MAX_INFLIGHT", 1024);
  inflight_ = (getenv("QCNN_COALESCE") != nullptr && atoi(getenv("QCNN_COALESCE")) == 0)
                  ? batchSize_ : static_cast<int>(all < cap ? all : cap);
  if (inflight_ < batchSize_) inflight_ = batchSize_;
  return buildDeviceModel();
}

// PrepFeatMap / PrepFeatBuf / PrepCtrdBuf / PrepAsmtBuf of the reference (src/CaffeEva.cc:328-623) happen
// on the device side of the C-ABI; here the loaded tensors are only handed over.  The object drives a device
// GROUP (include/qcnn_hip.h): by default every visible GPU; parameters go to rank 0 and are broadcast with RCCL.
bool CaffeEva::buildDeviceModel(void) {
  if (grp_ == nullptr) {
    std::vector<int> devs;                                 // empty = every visible device
    const char* one = getenv("QCNN_DEVICE");
    const char* many = getenv("QCNN_DEVICES");
    if (many != nullptr && *many != '\0' && std::string(many) != "all") {
      for (const char* q = many; *q != '\0';) {
        devs.push_back(atoi(q));
        while (*q != '\0' && *q != ',') ++q;
        if (*q == ',') ++q;
      }
    } else if (one != nullptr && *one != '\0') {
      devs.push_back(atoi(one));
    }
    if (qcnn_group_create(devs.empty() ? nullptr : devs.data(), static_cast<int>(devs.size()), &grp_) != 0) {
      grp_ = nullptr;
      return fail(std::string("cannot create the device group: ") + qcnn_group_last_error(nullptr));
    }
    ctx_ = qcnn_group_ctx(grp_, 0);
    printf("[INFO] device group: %d GPU(s)\n", qcnn_group_size(grp_));
  }
  const char* lut = getenv("QCNN_LUT");
  qcnn_group_set_option(grp_, QCNN_OPT_LUT_MODE, (lut && std::string(lut) == "exact") ? 0 : 1);
  // fast path by default (ReLU fused into the producing layer, first layer reads the batch in place, LRN + pool fused);
  // QCNN_KEEP_ALL=1 makes every layer write its own map, which is what GetFeatMap() needs for the fused ones
  const char* keep = getenv("QCNN_KEEP_ALL");
  keepAll_ = keep != nullptr && atoi(keep) != 0;
  qcnn_group_set_option(grp_, QCNN_OPT_KEEP_ALL, keepAll_ ? 1 : 0);
  qcnn_group_set_option(grp_, QCNN_OPT_PROFILE, 1);
  if (pinned_ == nullptr && pinnedCopy_ == nullptr) pinDataset();

  const int L = caffeParaObj.layerCnt;
  std::vector<QcnnLayerDesc> descs(L);
  for (int l = 0; l < L; ++l) descs[l] = toDesc(caffeParaObj.layerInfoLst[l]);
  if (qcnn_group_model_begin(grp_, L, descs.data(), caffeParaObj.imgChnIn, caffeParaObj.imgHeiIn, caffeParaObj.imgWidIn))
    return fail("qcnn_group_model_begin");
  for (int l = 0; l < L; ++l) {
    const ENUM_LyrType t = caffeParaObj.layerInfoLst[l].type;
    if (t != ENUM_LyrType::Conv && t != ENUM_LyrType::FCnt) continue;
    if (!enblAprx) {                     // precise path (Init(false)): dense kernels / weights, src/CaffeEva.cc:681-758, 932-966
      if (qcnn_group_model_set_layer_dense(grp_, l)) return fail("qcnn_group_model_set_layer_dense");
      continue;
    }
    const Matrix<float>& ctrd = caffeParaObj.layerParaLst[l].ctrdLst;   // [M][K][Cs]
    if (ctrd.GetDimCnt() != 3) return fail("layer without a 3-D ctrdLst");
    if (qcnn_group_model_set_layer_shape(grp_, l, ctrd.GetDimLen(0), ctrd.GetDimLen(1), ctrd.GetDimLen(2)))
      return fail("qcnn_group_model_set_layer_shape");
  }
  if (qcnn_group_model_commit(grp_, inflight_)) return fail("qcnn_group_model_commit");
  for (int l = 0; l < L; ++l) {
    const LayerInfo& li = caffeParaObj.layerInfoLst[l];
    if (li.type != ENUM_LyrType::Conv && li.type != ENUM_LyrType::FCnt) continue;
    const LayerPara& lp = caffeParaObj.layerParaLst[l];
    // the C-ABI trusts the declared (M, K, Cs) and the layer table: check what the files actually held
    int hwc[3];
    qcnn_fm_dims(ctx_, l + 1, hwc);
    const size_t Ct = static_cast<size_t>(hwc[2]);
    const size_t taps = (li.type == ENUM_LyrType::Conv) ? static_cast<size_t>(li.knlSiz) * li.knlSiz : 1;
    if (static_cast<size_t>(lp.biasVec.GetEleCnt()) != Ct)
      return fail("layer parameter files: biasVec does not have one entry per output channel");
    if (!enblAprx) {
      int in[3];
      qcnn_fm_dims(ctx_, l, in);
      const Matrix<float>& w = (li.type == ENUM_LyrType::Conv) ? lp.convKnlLst : lp.fcntWeiMat;
      const size_t want = (li.type == ENUM_LyrType::Conv) ? Ct * (static_cast<size_t>(in[2]) / li.grpCnt) * taps
                                                          : Ct * static_cast<size_t>(in[0]) * in[1] * in[2];
      if (static_cast<size_t>(w.GetEleCnt()) != want)
        return fail("layer parameter files: convKnl / fcntWei does not hold Ct x inputs x taps weights");
      if (qcnn_group_model_set_layer_weights(grp_, l, lp.biasVec.GetDataPtr(), w.GetDataPtr()))
        return fail("qcnn_group_model_set_layer_weights");
      continue;
    }
    const size_t M = static_cast<size_t>(lp.ctrdLst.GetDimLen(0));
    if (static_cast<size_t>(lp.asmtLst.GetEleCnt()) != Ct * taps * M)
      return fail("layer parameter files: asmtLst does not hold Ct x taps x M assignments");
    if (qcnn_group_model_set_layer_params(grp_, l, lp.biasVec.GetDataPtr(), lp.ctrdLst.GetDataPtr(), lp.asmtLst.GetDataPtr()))
      return fail("qcnn_group_model_set_layer_params");
  }
  float bcastMs = 0.0f;
  if (qcnn_group_model_broadcast(grp_, &bcastMs)) return fail("qcnn_group_model_broadcast");
  if (qcnn_group_size(grp_) > 1) printf("[INFO] parameters broadcast to %d GPUs with RCCL in %.3f ms\n", qcnn_group_size(grp_), bcastMs);
  // the reference prints the feature-map sizes here (src/CaffeEva.cc:403-410)
  for (int l = 0; l <= L; ++l) {
    int hwc[3];
    qcnn_fm_dims(ctx_, l, hwc);
    printf("layer #%2d: %4d x %4d x %4d x %4d (%6.2f MB)\n", l, batchSize_, hwc[0], hwc[1], hwc[2],
           batchSize_ * hwc[0] * hwc[1] * hwc[2] * 4 / 1024.0 / 1024.0);
  }
  // Warm-up (QCNN_WARMUP=0 skips it): two one-panel batches of zeros through the pipelined path, so that code objects,
  // staging buffers, streams and events exist before the first timed forward pass; the timers start from zero after it.
  const char* warm = getenv("QCNN_WARMUP");
  if (warm == nullptr || atoi(warm) != 0) {
    const int wn = inflight_ < 128 ? inflight_ : 128;
    const size_t perImg = static_cast<size_t>(caffeParaObj.imgChnIn) * caffeParaObj.imgHeiIn * caffeParaObj.imgWidIn;
    std::vector<float> zeros(perImg * wn, 0.0f);
    std::vector<uint16_t> t5(static_cast<size_t>(wn) * kLablCntPerData);
    const float* in[2] = {zeros.data(), zeros.data()};
    const int cnt[2] = {wn, wn};
    uint16_t* out[2] = {t5.data(), t5.data()};
    qcnn_group_set_option(grp_, QCNN_OPT_SMALL_BATCH, 0);
    if (qcnn_group_forward_host_batches(grp_, in, cnt, 2, nullptr, out)) return fail("warm-up forward pass");
    for (int r = 0; r < qcnn_group_size(grp_); ++r) qcnn_reset_layer_ms(qcnn_group_ctx(grp_, r));
  }
  modelReady_ = true;
  return true;
}

// The reference classifies batchCnt batches of batchSize images one after the other (src/CaffeEva.cc:151-211).
// Images are independent, so here the same images — same window rule, same prints, same results — are handed
// to the device group in chunks of up to `inflight_` images (QCNN_MAX_INFLIGHT), each chunk sharded over the
// GPUs: the reference's default of 100 batches of 1 image becomes ONE device batch of 100.
void CaffeEva::ExecForwardPass(void) {
  printf("[CHECK-POINT] entering CaffeEva::ExecForwardPass()\n");
  imagesDone_ = 0;
  if (!modelReady_) { fail("ExecForwardPass() before a successful LoadCaffePara()"); return; }
  if (dataLst.GetDimCnt() != 4) { fail("ExecForwardPass() before a successful LoadDataset()"); return; }
  const int dataCnt = dataLst.GetDimLen(0);
  const size_t perImg = static_cast<size_t>(dataLst.GetDimStp(0));
  // same values, pinned storage — for the leading pinnedImages_ images (pinDataset); a window beyond them is read from dataLst
  auto imageAt = [&](int first, int count) -> const float* {
    const float* base = (pinnedCopy_ != nullptr && first + count <= pinnedImages_) ? pinnedCopy_ : dataLst.GetDataPtr();
    return base + static_cast<size_t>(first) * perImg;
  };
  if (dataCnt < batchSize_) { fail("dataset smaller than one batch"); return; }
  lablVecPred.Create(dataCnt, kLablCntPerData, 1, 1);
  memset(lablVecPred.GetDataPtr(), 0, sizeof(uint16_t) * lablVecPred.GetEleCnt());
  const int batchesInData = (dataCnt + batchSize_ - 1) / batchSize_;
  std::vector<int> firstOf(batchCnt_);
  for (int b = 0; b < batchCnt_; ++b) {
    // same window rule as the reference (src/CaffeEva.cc:170-177): the last window is right-aligned
    int first = batchSize_ * b;
    if (b >= batchesInData - 1) first = dataCnt - batchSize_;
    if (first < 0 || first + batchSize_ > dataCnt) first = dataCnt - batchSize_;
    firstOf[b] = first;
  }
  // Device batches: chunks of up to `inflight_` images = whole logical batches.  All of them are handed over in ONE call,
  // so that the upload of a chunk runs under the layers of the one before it (qcnn_group_forward_host_batches).
  const int perChunk = inflight_ / batchSize_ > 0 ? inflight_ / batchSize_ : 1;      // logical batches per device batch
  const int chunks = (batchCnt_ + perChunk - 1) / perChunk;
  std::vector<const float*> in(chunks);
  std::vector<int> cnt(chunks);
  std::vector<uint16_t*> t5(chunks);
  std::vector<std::vector<float> > staging(chunks);
  std::vector<uint16_t> top5(static_cast<size_t>(batchCnt_) * batchSize_ * kLablCntPerData);
  for (int k = 0; k < chunks; ++k) {
    const int b0 = k * perChunk;
    const int nb = (b0 + perChunk <= batchCnt_) ? perChunk : batchCnt_ - b0;
    bool contiguous = true;
    for (int b = 1; b < nb; ++b) contiguous = contiguous && firstOf[b0 + b] == firstOf[b0 + b - 1] + batchSize_;
    in[k] = imageAt(firstOf[b0], nb * batchSize_);
    if (!contiguous) {                                   // right-aligned last window: gather the chunk
      staging[k].resize(static_cast<size_t>(nb) * batchSize_ * perImg);
      for (int b = 0; b < nb; ++b)
        memcpy(staging[k].data() + static_cast<size_t>(b) * batchSize_ * perImg,
               imageAt(firstOf[b0 + b], batchSize_), sizeof(float) * batchSize_ * perImg);
      in[k] = staging[k].data();
    }
    cnt[k] = nb * batchSize_;
    t5[k] = top5.data() + static_cast<size_t>(b0) * batchSize_ * kLablCntPerData;
  }
  for (int b = 0; b < batchCnt_; ++b) printf("processing the %d-th batch\n", b + 1);
  // every image through the panel kernels, whatever the chunking: coalesced and batch-by-batch runs give the same bits
  qcnn_group_set_option(grp_, QCNN_OPT_SMALL_BATCH, 0);
  swWall_.Resume();
  const bool ok = qcnn_group_forward_host_batches(grp_, in.data(), cnt.data(), chunks, nullptr, t5.data()) == 0;
  swWall_.Pause();
  if (!ok) { fail("qcnn_group_forward_host_batches"); return; }
  for (int b = 0; b < batchCnt_; ++b)
    for (int i = 0; i < batchSize_; ++i)
      for (int r = 0; r < kLablCntPerData; ++r)
        lablVecPred.SetEleAt(top5[(static_cast<size_t>(b) * batchSize_ + i) * kLablCntPerData + r], firstOf[b] + i, r, 0, 0);
  imagesDone_ = batchCnt_ * batchSize_;
}

void CaffeEva::ExecForwardPass(const Matrix<float>& imgDataIn, Matrix<float>* pProbVecOut) {
  printf("[CHECK-POINT] entering CaffeEva::ExecForwardPass()\n");
  if (!modelReady_) { fail("ExecForwardPass() before a successful LoadCaffePara()"); return; }
  int hwc[3];
  qcnn_fm_dims(ctx_, caffeParaObj.layerCnt, hwc);
  pProbVecOut->Resize(hwc[0] * hwc[1] * hwc[2]);
  memset(pProbVecOut->GetDataPtr(), 0, sizeof(float) * pProbVecOut->GetEleCnt());   // never hand back uninitialised memory
  lastError_.clear();
  qcnn_group_set_option(grp_, QCNN_OPT_SMALL_BATCH, 1);     // latency mode: the few-image kernels
  swWall_.Resume();
  if (qcnn_forward_host(ctx_, imgDataIn.GetDataPtr(), 1, pProbVecOut->GetDataPtr(), nullptr) != 0)
    fail("qcnn_forward_host");                                                      // GetErrorMsg() is non-empty: callers check it
  swWall_.Pause();
}

bool CaffeEva::GetFeatMap(const int layerInd, const int dataCnt, Matrix<float>* pFeatMap) {
  if (!modelReady_) return fail("GetFeatMap() before a successful LoadCaffePara()");
  int hwc[3];
  if (qcnn_fm_dims(ctx_, layerInd, hwc)) return fail("qcnn_fm_dims");
  pFeatMap->Create(dataCnt, hwc[0], hwc[1], hwc[2]);
  if (qcnn_get_layer_output(ctx_, layerInd, dataCnt, pFeatMap->GetDataPtr()))
    return fail(keepAll_ ? "qcnn_get_layer_output" : "qcnn_get_layer_output (maps the fast path fuses away need QCNN_KEEP_ALL=1)");
  return true;
}

void CaffeEva::CalcPredAccu(void) {
  printf("[CHECK-POINT] entering CaffeEva::CalcPredAccu()\n");
  const int dataCnt = imagesDone_;
  if (dataCnt <= 0 || lablVecGrth.GetEleCnt() < dataCnt) {
    printf("[ERROR] no predictions to score\n");
    return;
  }
  // cumulative top-k hits over the images that were classified (src/CaffeEva.cc:274-294 scores the first
  // kDataCntInBatch * kBatchCntProc entries; with the defaults that is the same 100 images)
  unsigned hits[kLablCntPerData] = {0, 0, 0, 0, 0};
  const uint16_t* truth = lablVecGrth.GetDataPtr();
  for (int i = 0; i < dataCnt; ++i)
    for (int r = 0; r < kLablCntPerData; ++r)
      if (lablVecPred.GetEleAt(i, r, 0, 0) == truth[i]) hits[r]++;
  unsigned acc = 0;
  for (int r = 0; r < kLablCntPerData; ++r) {
    acc += hits[r];
    printf("ACCURACY@%d: %d, %.2f%%\n", r + 1, acc, 100.0 * acc / dataCnt);
  }
}

float CaffeEva::DispElpsTime(void) {
  const int L = caffeParaObj.layerCnt;
  // HIP-event time of every launch of every forward pass since the last Init(), summed per layer (rank 0 of the
  // device group; the ranks run concurrently on equal shares)
  std::vector<double> ms(L > 0 ? L : 1, 0.0);
  int forwards = 0;
  if (modelReady_) qcnn_get_layer_total_ms(ctx_, ms.data(), nullptr, &forwards);
  double byType[7] = {0, 0, 0, 0, 0, 0, 0};
  double convK = 0.0, fcK = 0.0, total = 0.0;
  for (int l = 0; l < L; ++l) {
    const double s = ms[l] * 1e-3;                     // seconds over all recorded forward passes
    const int t = static_cast<int>(caffeParaObj.layerInfoLst[l].type);
    byType[t] += s;
    total += s;
    if (t == static_cast<int>(ENUM_LyrType::Conv)) convK += s;
    if (t == static_cast<int>(ENUM_LyrType::FCnt)) fcK += s;
  }
  printf("swAllLayers: %.4f (s)\n", total);
  printf("swConvLayer: %.4f (s)\n", byType[static_cast<int>(ENUM_LyrType::Conv)]);
  printf("swPoolLayer: %.4f (s)\n", byType[static_cast<int>(ENUM_LyrType::Pool)]);
  printf("swFCntLayer: %.4f (s)\n", byType[static_cast<int>(ENUM_LyrType::FCnt)]);
  printf("swReLuLayer: %.4f (s)\n", byType[static_cast<int>(ENUM_LyrType::ReLU)]);
  printf("swLoRNLayer: %.4f (s)\n", byType[static_cast<int>(ENUM_LyrType::LoRN)]);
  printf("swDrptLayer: %.4f (s)\n", byType[static_cast<int>(ENUM_LyrType::Drpt)]);
  printf("swSMaxLayer: %.4f (s)\n", byType[static_cast<int>(ENUM_LyrType::SMax)]);
  printf("swCompLkupTblConv: %.4f (s)\n", 0.0);        // fused into the look-up kernel
  printf("swEstiInPdValConv: %.4f (s)\n", convK);
  printf("swCompLkupTblFCnt: %.4f (s)\n", 0.0);
  printf("swEstiInPdValFCnt: %.4f (s)\n", fcK);
  printf("swDebugTimePri: %.4f (s)\n", static_cast<double>(swWall_.GetTime()));   // host wall clock incl. H2D/D2H
  printf("swDebugTimeSec: %.4f (s)\n", 0.0);
  for (int l = 0; l < L; ++l) printf("swIndvLayerLst #%2d: %.4f (s)\n", l + 1, ms[l] * 1e-3);
  Init(enblAprx);
  return static_cast<float>(total);
}
