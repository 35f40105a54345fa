// qcnn_group.hip — several MI355X behind the C-ABI (include/qcnn_hip.h, "device group" section).
//
// The reference classifies its images one after the other in a single loop (src/CaffeEva.cc:151-211); images
// are independent, so a batch shards over the GPUs of a node without any exchange during the forward pass
// (SURVEY.md §8e).  A QcnnGroup owns one QcnnCtx per device and one RCCL communicator over them (single
// process, ncclCommInitAll).  The only collective is the one-time ncclBroadcast of rank 0's packed parameter
// arena (biases, permuted codebooks, row-offset tables) over xGMI at load time; a forward pass is one host
// thread per GPU, each running its contiguous block of the batch on its own context and stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/qcnn_hip.h"

struct QcnnGroup {
  std::vector<int> devs;
  std::vector<QcnnCtx*> ctx;
  std::vector<ncclComm_t> comm;
  std::string err;
  int classes = 0;
  size_t inElems = 0;
  float bcastMs = 0.0f;
  unsigned long long arenaSum[2] = {0, 0};   // checksum every rank's arena agreed on at the last broadcast (qcnn_group_arena_checksum)
  bool broadcastDone = false;
  bool dupDevices = false;      // QCNN_GROUP_ALLOW_DUP: several ranks on one device, no RCCL communicator
  int smallBatch = 1;           // QCNN_OPT_SMALL_BATCH as the caller set it; applied per forward by the GLOBAL batch size
};

namespace {

std::string g_groupCreateError;

int gfail(QcnnGroup* g, const char* fmt, ...) {
  char buf[768];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (g) g->err = buf; else g_groupCreateError = buf;
  return 1;
}

// contiguous block of rank r: image i goes to rank i * G / n (the same rule as dist.shard_bounds)
void shard(int n, int r, int G, int* first, int* count) {
  const long long lo = (long long)n * r / G, hi = (long long)n * (r + 1) / G;
  *first = (int)lo;
  *count = (int)(hi - lo);
}

#define FOR_ALL(g, call)                                                                                \
  do {                                                                                                  \
    for (size_t r_ = 0; r_ < (g)->ctx.size(); ++r_) {                                                   \
      QcnnCtx* c = (g)->ctx[r_];                                                                        \
      if (call) return gfail((g), "rank %zu (device %d): %s", r_, (g)->devs[r_], qcnn_last_error(c));   \
    }                                                                                                   \
  } while (0)

}  // namespace

extern "C" {

const char* qcnn_group_last_error(const QcnnGroup* g) { return g ? g->err.c_str() : g_groupCreateError.c_str(); }

int qcnn_group_create(const int* device_ids, int n_dev, QcnnGroup** out) {
  if (!out) return gfail(nullptr, "qcnn_group_create: out == NULL");
  *out = nullptr;
  int visible = 0;
  if (qcnn_device_count(&visible) || visible <= 0)
    return gfail(nullptr, "no HIP device available (%s); this library has no CPU path", qcnn_last_error(nullptr));
  QcnnGroup* g = new QcnnGroup;
  if (!device_ids || n_dev <= 0) {
    for (int d = 0; d < visible; ++d) g->devs.push_back(d);
  } else {
    const char* dup = getenv("QCNN_GROUP_ALLOW_DUP");
    const bool allowDup = dup != nullptr && atoi(dup) != 0;
    for (int i = 0; i < n_dev; ++i) {
      for (int j = 0; j < i; ++j)
        if (device_ids[j] == device_ids[i]) {
          if (!allowDup) { delete g; return gfail(nullptr, "device %d listed twice", device_ids[i]); }
          g->dupDevices = true;
        }
      g->devs.push_back(device_ids[i]);
    }
  }
  for (int d : g->devs) {
    QcnnCtx* c = nullptr;
    if (qcnn_ctx_create(d, nullptr, &c)) {
      gfail(nullptr, "device %d: %s", d, qcnn_last_error(nullptr));
      for (QcnnCtx* k : g->ctx) qcnn_ctx_destroy(k);
      delete g;
      return 1;
    }
    g->ctx.push_back(c);
  }
  g->comm.assign(g->devs.size(), nullptr);
  const ncclResult_t nr = g->dupDevices ? ncclSuccess
                                        : ncclCommInitAll(g->comm.data(), (int)g->devs.size(), g->devs.data());
  if (nr != ncclSuccess) {
    gfail(nullptr, "ncclCommInitAll over %zu device(s) -> %s", g->devs.size(), ncclGetErrorString(nr));
    for (QcnnCtx* k : g->ctx) qcnn_ctx_destroy(k);
    delete g;
    return 1;
  }
  *out = g;
  return 0;
}

int qcnn_group_destroy(QcnnGroup* g) {
  if (!g) return 0;
  for (size_t r = 0; r < g->comm.size(); ++r)
    if (g->comm[r]) { (void)hipSetDevice(g->devs[r]); (void)ncclCommDestroy(g->comm[r]); }
  for (QcnnCtx* c : g->ctx) qcnn_ctx_destroy(c);
  delete g;
  return 0;
}

int qcnn_group_size(const QcnnGroup* g) { return (int)g->ctx.size(); }

QcnnCtx* qcnn_group_ctx(QcnnGroup* g, int rank) {
  return (rank >= 0 && rank < (int)g->ctx.size()) ? g->ctx[rank] : nullptr;
}

int qcnn_group_shard_bounds(const QcnnGroup* g, int n, int rank, int* first, int* count) {
  if (rank < 0 || rank >= (int)g->ctx.size() || n < 0) return 1;
  shard(n, rank, (int)g->ctx.size(), first, count);
  return 0;
}

int qcnn_group_set_option(QcnnGroup* g, int option, int value) {
  if (option == QCNN_OPT_SMALL_BATCH) g->smallBatch = value ? 1 : 0;
  FOR_ALL(g, qcnn_set_option(c, option, value));
  return 0;
}

int qcnn_group_model_begin(QcnnGroup* g, int layer_cnt, const QcnnLayerDesc* layers, int in_c, int in_h, int in_w) {
  g->broadcastDone = false;
  FOR_ALL(g, qcnn_model_begin(c, layer_cnt, layers, in_c, in_h, in_w));
  int hwc[3];
  if (qcnn_fm_dims(g->ctx[0], layer_cnt, hwc)) return gfail(g, "%s", qcnn_last_error(g->ctx[0]));
  g->classes = hwc[0] * hwc[1] * hwc[2];
  g->inElems = (size_t)in_c * in_h * in_w;
  return 0;
}

int qcnn_group_model_set_layer_shape(QcnnGroup* g, int layer, int M, int K, int Cs) {
  FOR_ALL(g, qcnn_model_set_layer_shape(c, layer, M, K, Cs));
  return 0;
}

int qcnn_group_model_set_layer_dense(QcnnGroup* g, int layer) {
  FOR_ALL(g, qcnn_model_set_layer_dense(c, layer));
  return 0;
}

int qcnn_group_model_set_layer_weights(QcnnGroup* g, int layer, const float* bias, const float* weights_file) {
  if (qcnn_model_set_layer_weights(g->ctx[0], layer, bias, weights_file))
    return gfail(g, "rank 0 (device %d): %s", g->devs[0], qcnn_last_error(g->ctx[0]));
  g->broadcastDone = false;
  return 0;
}

int qcnn_group_model_commit(QcnnGroup* g, int max_batch) {
  if (max_batch <= 0) return gfail(g, "max_batch must be positive");
  const int G = (int)g->ctx.size();
  const int share = (max_batch + G - 1) / G;          // the largest block shard_bounds can hand to a rank
  FOR_ALL(g, qcnn_model_commit(c, share, nullptr));
  return 0;
}

int qcnn_group_model_set_layer_params(QcnnGroup* g, int layer, const float* bias, const float* ctrd_file,
                                      const uint8_t* asmt_file) {
  if (qcnn_model_set_layer_params(g->ctx[0], layer, bias, ctrd_file, asmt_file))
    return gfail(g, "rank 0 (device %d): %s", g->devs[0], qcnn_last_error(g->ctx[0]));
  g->broadcastDone = false;
  return 0;
}

int qcnn_group_model_broadcast(QcnnGroup* g, float* elapsed_ms) {
  const int G = (int)g->ctx.size();
  std::vector<void*> arena(G, nullptr);
  size_t bytes = 0;
  for (int r = 0; r < G; ++r) {
    size_t b = 0;
    if (qcnn_model_arena_ptr(g->ctx[r], &arena[r], &b)) return gfail(g, "rank %d: %s", r, qcnn_last_error(g->ctx[r]));
    if (r && b != bytes) return gfail(g, "rank %d plans a %zu-byte arena, rank 0 %zu", r, b, bytes);
    bytes = b;
  }
  for (int r = 0; r < G; ++r)
    if (qcnn_sync(g->ctx[r])) return gfail(g, "rank %d: %s", r, qcnn_last_error(g->ctx[r]));
  const auto t0 = std::chrono::steady_clock::now();
  if (g->dupDevices) {
    // test rig (QCNN_GROUP_ALLOW_DUP): ranks share devices, RCCL cannot span them — copy rank 0's arena instead
    for (int r = 1; r < G; ++r) {
      (void)hipSetDevice(g->devs[r]);
      hipStream_t st = static_cast<hipStream_t>(qcnn_ctx_stream(g->ctx[r]));
      const hipError_t e = (g->devs[r] == g->devs[0])
                               ? hipMemcpyAsync(arena[r], arena[0], bytes, hipMemcpyDeviceToDevice, st)
                               : hipMemcpyPeerAsync(arena[r], g->devs[r], arena[0], g->devs[0], bytes, st);
      if (e != hipSuccess) return gfail(g, "arena copy to rank %d -> %s", r, hipGetErrorString(e));
    }
  } else {
    ncclResult_t nr = ncclGroupStart();
    for (int r = 0; r < G && nr == ncclSuccess; ++r) {
      (void)hipSetDevice(g->devs[r]);
      nr = ncclBroadcast(arena[r], arena[r], bytes, ncclUint8, 0, g->comm[r],
                         static_cast<hipStream_t>(qcnn_ctx_stream(g->ctx[r])));
    }
    const ncclResult_t ne = ncclGroupEnd();
    if (nr != ncclSuccess || ne != ncclSuccess)
      return gfail(g, "ncclBroadcast of the %zu-byte parameter arena -> %s", bytes,
                   ncclGetErrorString(nr != ncclSuccess ? nr : ne));
  }
  for (int r = 0; r < G; ++r)
    if (qcnn_sync(g->ctx[r])) return gfail(g, "rank %d: %s", r, qcnn_last_error(g->ctx[r]));
  g->bcastMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // every rank's arena must now hold rank 0's bytes: compare device-side checksums before any rank is declared loaded (a
  // broadcast that silently moved nothing — or the wrong bytes — would otherwise surface as wrong class scores on some shards)
  unsigned long long want[2] = {0, 0};
  for (int r = 0; r < G; ++r) {
    unsigned long long got[2] = {0, 0};
    if (qcnn_model_arena_checksum(g->ctx[r], got)) return gfail(g, "rank %d: %s", r, qcnn_last_error(g->ctx[r]));
    if (r == 0) { want[0] = got[0]; want[1] = got[1]; }
    else if (got[0] != want[0] || got[1] != want[1])
      return gfail(g, "parameter broadcast: rank %d (device %d) holds arena checksum %016llx:%016llx, rank 0 %016llx:%016llx", r, g->devs[r],
                   got[0], got[1], want[0], want[1]);
  }
  g->arenaSum[0] = want[0]; g->arenaSum[1] = want[1];
  for (int r = 1; r < G; ++r)
    if (qcnn_model_mark_loaded(g->ctx[r])) return gfail(g, "rank %d: %s", r, qcnn_last_error(g->ctx[r]));
  g->broadcastDone = true;
  if (elapsed_ms) *elapsed_ms = g->bcastMs;
  return 0;
}

int qcnn_group_forward_host_batches(QcnnGroup* g, const float* const* in_host, const int* n, int nb,
                                    float* const* prob_host, uint16_t* const* top5_host) {
  const int G = (int)g->ctx.size();
  if (nb <= 0 || !in_host || !n) return gfail(g, "no batches");
  int nMax = 0;
  for (int b = 0; b < nb; ++b) {
    if (n[b] <= 0 || !in_host[b]) return gfail(g, "batch %d: %d images", b, n[b]);
    nMax = n[b] > nMax ? n[b] : nMax;
  }
  if (G > 1 && !g->broadcastDone) return gfail(g, "qcnn_group_model_broadcast must follow the parameter upload");
  // The kernel family follows the GLOBAL batch, not the shard: an image must not change bits with the number of GPUs
  // (a shard of one or two images of a larger batch would otherwise take the few-image kernels).
  for (int r = 0; r < G; ++r)
    if (qcnn_set_option(g->ctx[r], QCNN_OPT_SMALL_BATCH, (g->smallBatch && nMax <= QCNN_SMALL_BATCH_MAX) ? 1 : 0))
      return gfail(g, "rank %d: %s", r, qcnn_last_error(g->ctx[r]));
  std::vector<int> rc(G, 0);
  auto work = [&](int r) {
    std::vector<const float*> in;
    std::vector<int> cnt;
    std::vector<float*> pr;
    std::vector<uint16_t*> t5;
    for (int b = 0; b < nb; ++b) {                    // this rank's block of every batch; empty blocks are skipped
      int first = 0, count = 0;
      shard(n[b], r, G, &first, &count);
      if (count == 0) continue;
      in.push_back(in_host[b] + (size_t)first * g->inElems);
      cnt.push_back(count);
      pr.push_back((prob_host && prob_host[b]) ? prob_host[b] + (size_t)first * g->classes : nullptr);
      t5.push_back((top5_host && top5_host[b]) ? top5_host[b] + (size_t)first * 5 : nullptr);
    }
    if (in.empty()) return;
    rc[r] = (in.size() == 1) ? qcnn_forward_host(g->ctx[r], in[0], cnt[0], pr[0], t5[0])
                             : qcnn_forward_host_batches(g->ctx[r], in.data(), cnt.data(), (int)in.size(), pr.data(), t5.data());
  };
  if (G == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int r = 1; r < G; ++r) th.emplace_back(work, r);
    work(0);
    for (std::thread& t : th) t.join();
  }
  for (int r = 0; r < G; ++r)
    if (rc[r]) return gfail(g, "rank %d (device %d): %s", r, g->devs[r], qcnn_last_error(g->ctx[r]));
  return 0;
}

int qcnn_group_forward_host(QcnnGroup* g, const float* in_nchw_host, int n, float* prob_host, uint16_t* top5_host) {
  if (n <= 0) return gfail(g, "batch %d must be positive", n);
  return qcnn_group_forward_host_batches(g, &in_nchw_host, &n, 1, &prob_host, &top5_host);
}

// Device-resident form: rank r's block of the batch (qcnn_group_shard_bounds) already lies on rank r's device; every rank's
// layers are ENQUEUED on its own stream (no host thread, no PCIe), qcnn_group_sync waits for all of them.
int qcnn_group_forward(QcnnGroup* g, const float* const* in_dev, int n, float* const* prob_dev, uint16_t* const* top5_dev) {
  const int G = (int)g->ctx.size();
  if (!in_dev) return gfail(g, "qcnn_group_forward: in_dev == NULL (one device pointer per rank)");
  if (n <= 0) return gfail(g, "batch %d must be positive", n);
  if (G > 1 && !g->broadcastDone) return gfail(g, "qcnn_group_model_broadcast must follow the parameter upload");
  for (int r = 0; r < G; ++r) {
    int first = 0, count = 0;
    shard(n, r, G, &first, &count);
    if (count == 0) continue;
    if (!in_dev[r]) return gfail(g, "rank %d: no input pointer for its %d images", r, count);
    if (qcnn_set_option(g->ctx[r], QCNN_OPT_SMALL_BATCH, (g->smallBatch && n <= QCNN_SMALL_BATCH_MAX) ? 1 : 0) ||
        qcnn_forward(g->ctx[r], in_dev[r], count, prob_dev ? prob_dev[r] : nullptr, top5_dev ? top5_dev[r] : nullptr))
      return gfail(g, "rank %d (device %d): %s", r, g->devs[r], qcnn_last_error(g->ctx[r]));
  }
  return 0;
}

int qcnn_group_sync(QcnnGroup* g) {
  FOR_ALL(g, qcnn_sync(c));
  return 0;
}

}  // extern "C"

/* Checksum pair every rank's arena agreed on at the last qcnn_group_model_broadcast (qcnn_model_arena_checksum). */
int qcnn_group_arena_checksum(QcnnGroup* g, unsigned long long* sum2) {
  if (!g->broadcastDone) return gfail(g, "qcnn_group_model_broadcast has not run");
  if (sum2) { sum2[0] = g->arenaSum[0]; sum2[1] = g->arenaSum[1]; }
  return 0;
}
