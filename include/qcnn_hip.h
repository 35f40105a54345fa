/* qcnn_hip.h — C-ABI of the MI355X (gfx950) Quantized-CNN approximate forward pass.
 *
 * This is the drop-in boundary: plain C types, caller-owned host/device buffers,
 * library-owned activations, 0 = OK / non-zero = error + qcnn_last_error().  The host side of this
 * repository (include/CaffeEva.h, quantized-cnn_amd/host/caffe_eva.cc) binds exactly these entry
 * points where the reference's CaffeEva calls its CPU routines; INTEGRATION.md shows the same
 * binding applied to the reference tree.  Built by `hipcc --offload-arch=gfx950` into
 * quantized-cnn_amd/libqcnn_hip.so.  There is no CPU fallback: without a HIP device every compute
 * entry point fails with an error string.
 *
 * Reference interface each entry point replaces (file:line in CAS-CLab/quantized-cnn):
 *   qcnn_model_begin            CaffePara::ConfigLayer_* tables            src/CaffePara.cc:20-237
 *                               + CaffeEva::PrepFeatMap size rule          src/CaffeEva.cc:328-411
 *   qcnn_model_set_layer_params CaffePara::LoadLayerPara result ->         src/CaffePara.cc:239-306
 *                               CaffeEva::PrepCtrdBuf / PrepAsmtBuf        src/CaffeEva.cc:534-623
 *   qcnn_model_set_layer_params_cbn  FileIO::ReadCbnFile decode + PrepAsmtBuf   include/FileIO.h:109-178
 *   qcnn_model_commit           CaffeEva::PrepFeatBuf (buffer planning)    src/CaffeEva.cc:413-532
 *   qcnn_forward[_host]         CaffeEva::ExecForwardPass layer loop       src/CaffeEva.cc:151-261
 *                               = CalcFeatMap dispatcher                   src/CaffeEva.cc:625-670
 *                               -> CalcFeatMap_ConvAprx                    src/CaffeEva.cc:760-868
 *                               -> CalcFeatMap_FCntAprx                    src/CaffeEva.cc:968-1025
 *                               -> GetInPdMat (look-up-table build)        src/CaffeEva.cc:1261-1296
 *                               -> _ReLu/_LoRN/_Pool/_Drpt/_SMax           src/CaffeEva.cc:870-921,1027-1116
 *                               + CvtFeatMapToLablVec (top-5)              src/CaffeEva.cc:1162-1190
 *   qcnn_forward_host_batches   the batch loop itself (one batch after the  src/CaffeEva.cc:168-206
 *                               other), uploads overlapped with compute
 *   qcnn_host_register          pins CaffeEva::dataLst (LoadDataset)       src/CaffeEva.cc:95-107
 *   qcnn_forward_u8             BmpImgIO::RmMeanImg + CropImg in front     src/BmpImgIO.cc:180-224
 *   qcnn_model_set_layer_dense  CaffePara::LoadLayerPara(false, ..) result src/CaffePara.cc:290-302
 *   / _set_layer_weights        -> CalcFeatMap_ConvPrec / _FCntPrec        src/CaffeEva.cc:681-758, 932-966
 *   qcnn_run_layer              CaffeEva::CalcFeatMap on one layer         src/CaffeEva.cc:625-670
 *   qcnn_get_layer_output       featMapLst[l] read-back (parity dumps)     include/CaffeEva.h:109
 *   qcnn_get_layer_ms           swIndvLayerLst / DispElpsTime              src/CaffeEva.cc:297-326
 *   qcnn_group_*                the image loop of ExecForwardPass(void)    src/CaffeEva.cc:151-211
 *                               spread over the GPUs of one node (the reference has one loop on one core)
 */
#ifndef QCNN_HIP_H_
#define QCNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QCNN_ABI_VERSION 5   /* 5 (round 6): QCNN_OPT_HALF8 (half-panel eight-wave workgroups), qcnn_model_arena_checksum / qcnn_group_arena_checksum,
                              * qcnn_model_mark_loaded drops the lazily built fp16 program tables.  4 (round 5): QCNN_OPT_LUT_MODE 2 = fp16 table storage, 3 = fp16 tables + fp16 sums (the bf16-pair builder
                              * of version 3 is gone); qcnn_set_option rejects out-of-range values of QCNN_OPT_SYM / _SLIDE / _SYM8;
                              * qcnn_model_set_layer_shape accepts up to 256 code words per sub-space; qcnn_group_forward */

#define QCNN_SMALL_BATCH_MAX 3   /* batches up to this size can take the few-image kernels (QCNN_OPT_SMALL_BATCH) */

typedef struct QcnnCtx QcnnCtx;

/* layer type codes: the order of ENUM_LyrType (include/CaffePara.h:26) */
enum { QCNN_CONV = 0, QCNN_POOL = 1, QCNN_FCNT = 2, QCNN_RELU = 3, QCNN_LORN = 4, QCNN_DRPT = 5, QCNN_SMAX = 6 };

/* same fields as the reference's LayerInfo (include/CaffePara.h:28-44) */
typedef struct {
  int type;
  int padSiz, knlSiz, knlCnt, grpCnt, stride, nodCnt, lrnSiz;
  float lrnAlp, lrnBet, lrnIni, drpRat;
} QcnnLayerDesc;

/* options for qcnn_set_option */
enum {
  QCNN_OPT_LUT_MODE = 0,   /* 0 = exact (VALU mul+add in the reference's order: conv/FC outputs are
                              bit-identical to the reference's -O2 native build); 1 = MFMA
                              (v_mfma_f32_16x16x4_f32, fused multiply-add chain; default); 2 = fp16 LUT STORAGE
                              (BASELINE.json configs[4], opt-in study: outside the 1e-4 bar): every table entry is
                              rounded to fp16 when it is stored, sums stay fp32 — conv layers with K = 128 and complete
                              4- / 8-dim sub-spaces and FC layers with 32 code words of 4 dims keep half-size tables in
                              LDS (256-byte rows, ds_read_b64 look-ups: k_conv_sym8 / k_fc_sym8 in their fp16 form),
                              every other layer rounds the entries and keeps them in f32 slots (same values); 3 = as 2 with the
                              RUNNING SUMS of those layers in packed fp16 too (rounded after every addition; the accumulate
                              half of the configs[4] study, 1e-2 territory): half the accumulator registers, so a
                              workgroup's tile is twice as large and every source pixel's table is built a third less often */
  QCNN_OPT_KEEP_ALL = 1,   /* 1 = every layer writes its own feature map (layer-for-layer dumps, default);
                              0 = fast path: ReLU fused into the producing conv/FC epilogue, the first conv layer
                              reads the NCHW input in place, and an LRN layer followed by a 3x3 / stride 2 / pad 0
                              max-pool runs as one kernel with it once a sub-batch is large enough to fill the chip
                              (the normalised map then does not exist: qcnn_get_layer_output fails for it) */
  QCNN_OPT_PROFILE = 2,    /* 1 = bracket every layer launch with HIP events (qcnn_get_layer_ms) */
  QCNN_OPT_SMALL_BATCH = 4, /* 1 (default): batches of up to QCNN_SMALL_BATCH_MAX images run the conv/FC layers with the few-image kernels
                              (lanes = output channels; one image no longer costs a 128-image panel).  Their sums
                              run over sub-space chunks first: equal to the panel kernels to rounding (~1e-6), not bit
                              for bit; 0 = panel kernels for every batch size (batch-size-invariant bits).  The exact
                              builder (QCNN_OPT_LUT_MODE = 0) always uses the panel kernels. */
  QCNN_OPT_SPLIT = 5,      /* 1 (default): a conv/FC launch that cannot fill the 256 CUs with whole tiles (a few panels, i.e.
                              one GPU's share of a batch sharded over the GPUs of a node) cuts the look-up sequences of the
                              tail of its tiles into slices run by separate workgroups and adds the partial sums in a
                              fixed order; FC layers pick their sub-space split for the panel count of the launch.  Results
                              then depend on the batch size to rounding (~1e-6).  Never applied with the exact builder
                              (QCNN_OPT_LUT_MODE = 0).  0 = one workgroup per tile, batch-size-invariant bits. */
  QCNN_OPT_SLIDE = 7,      /* 1 (default): conv layers with K = 128 whose windows overlap in at most three output rows
                              (knl <= 3 * stride) and whose channel count leaves room for three accumulator slots may run
                              the SLIDING kernel — a workgroup sweeps the source rows under a segment of one output column
                              and builds every source pixel of the strip once — when the launch planner predicts it to be
                              faster than the tile kernel.  Same summation order as the tile kernels ((kh, kw, m)), MFMA
                              builders only.  0 = tile kernels only; 2 = whenever
                              a layer is eligible, whatever the planner predicts (tests). */
  QCNN_OPT_DECODE = 8,     /* 1 (default): a conv layer with ONE sub-space of <= 4 dims (the RGB input layer: AlexNet conv1,
                              VGG-16 conv1_1) is evaluated through the code words its assignments name — the look-up
                              tables of src/CaffeEva.cc:1261-1296 are not materialised, the same sum goes to the matrix
                              pipe (a look-up there stands for <= 4 multiply-adds and the table of a pixel costs more than
                              the look-ups it serves).  Same parameters, same function, results within the MFMA modes'
                              tolerance; the network input is then packed into panels first.  MFMA modes only
                              (QCNN_OPT_LUT_MODE 1), batches above QCNN_SMALL_BATCH_MAX.  0 = table kernels for every layer */
  QCNN_OPT_SYM = 9,        /* 1 (default): a conv layer with exactly 128 channels per group and K = 128 (AlexNet conv2) that neither
                              slides nor splits may run SYMMETRIC workgroups — all 16 waves build and gather, 8 channels x a
                              2x2 tile per wave: fewer table builds per output position and no idle gather lane — when the
                              launch planner predicts them faster.  Bit-identical to the tile kernels; f32 MFMA mode only.
                              0 = never; 2 = whenever eligible (tests) */
  QCNN_OPT_SYM8 = 10,      /* 1 (default): a conv layer with K = 128, complete 4- or 8-dim sub-spaces and more than 64 channels per
                              group may run EIGHT-wave symmetric workgroups with 256 registers per wave (k_conv_sym8): twice the
                              accumulators of the 16-wave kernels per CU, i.e. larger output tiles (384 channels x 1x2, 256 x 1x3,
                              192 x 2x2, 128 x 2x3) and a third fewer table builds per output position — when the launch planner
                              predicts them faster than the tile / sliding / 16-wave symmetric launch.  Bit-identical to the tile
                              kernels; f32 MFMA mode only.  The same workgroups also SLIDE (segments of strips of one or two output
                              columns with 3 or 5 accumulator slots: 3x3 / 1 and 5x5 / 1 layers of up to 256 channels per workgroup build
                              every source pixel of a strip once: 3 or 2 table builds per output position instead of 4 - 5) where the
                              planner predicts that faster: VGG-16's 256- / 512-channel layers.  0 = never; 2 = tile form whenever
                              eligible, 3 = sliding form whenever eligible (tests); other values are refused.  qcnn_get_layer_split reports
                              (-5, slices per tile: QCNN_OPT_SPLIT cuts its tiles too) for the tile form, (-6, segments per column) for the
                              sliding form.  FC layers with 32 code words of 4 dims
                              (AlexNet / VGG-16 fc6, fc7) run the same eight waves (k_fc_sym8: 96 channels per wave, offsets through
                              LDS-DMA, software-pipelined look-ups) unless the option is 0 */
  QCNN_OPT_PACKED_FC = 11, /* 1: for batches of up to QCNN_SMALL_BATCH_MAX images the FC layers read their assignments from the
                              BIT-PACKED stream the reference's .cbn files hold (4 / 5 bits per assignment, file order [Ct][M], kept
                              resident in the arena beside the one-byte table of the panel kernels) and unpack them in the kernel:
                              AlexNet streams 10.5 MB of FC assignments per image instead of 16.8 MB of bytes (of which the strided
                              byte reads fetched 64-byte lines: 4x).  Same bits as the byte path.  0 (default) = bytes: the unpack arithmetic of the
                              block-structured stream costs more than the traffic it saves (fc6 0.056 against 0.035 ms at one image) */
  QCNN_OPT_DIRECT_DEC = 12, /* 1 (default): on the fast path (QCNN_OPT_KEEP_ALL = 0) an unpadded first conv layer that runs through its
                              decoded code words (QCNN_OPT_DECODE) reads the caller's NCHW batch in place (k_conv_dec_nchw: k over the
                              columns of a kernel row) — no pack pass into panels (AlexNet: 1.24 GB of HBM traffic per 1000 images).
                              0 = pack, then the panel kernel.  qcnn_get_layer_split reports (-3, 2) against (-3, 1) */
  QCNN_OPT_HALF8 = 13,     /* 1 (default): a conv layer with K = 128, complete 4- or 8-dim sub-spaces and 128 / 192 / 256 / 384 / 512
                              output channels per group (or chunks of 512) may run in eight-wave workgroups of HALF a panel (64 images:
                              twice the output tile per workgroup, half the table build per stage — qcnn_half8.hip) where the launch
                              planner predicts them faster than every other form — in their tile form or, for windows that overlap in
                              three output rows, their SLIDING form (segments of strips of output columns, every source pixel built once
                              per segment); 2 = the tile form whenever eligible, 3 = the sliding form whenever eligible (tests); 0 = never.
                              Same table entries in the same order: bit-identical to the other f32 table kernels.
                              qcnn_get_layer_split reports (-9, 1) / (-10, segments per column) */
  QCNN_OPT_HOST_CHUNK = 6, /* panels per chunk (default 2) of a qcnn_forward_host batch of at least two chunks: every chunk is
                              uploaded on a copy stream and its layers start when it has arrived, so the upload of chunk
                              k + 1 runs under the layers of chunk k; all chunks fill the same whole-batch feature maps.
                              0 = always one launch per layer for the whole batch */
  QCNN_OPT_STREAMS = 3     /* 1..4 (default 2): a forward is cut into that many sub-batches of whole 128-image
                              panels which run concurrently on separate HIP streams (LDS-bound conv/FC kernels
                              of one overlap HBM-bound glue kernels of another); results do not depend on it */
};

/* ---- context ---- */
/* device_id: HIP device ordinal.  stream: a hipStream_t to enqueue on (e.g. torch's current stream),
 * or NULL to let the library create its own. */
int qcnn_ctx_create(int device_id, void* stream, QcnnCtx** out);
int qcnn_ctx_destroy(QcnnCtx* ctx);
/* last error of ctx (or of ctx creation when ctx == NULL); never NULL */
const char* qcnn_last_error(const QcnnCtx* ctx);
int qcnn_abi_version(void);
int qcnn_device_count(int* count);
int qcnn_set_option(QcnnCtx* ctx, int option, int value);
int qcnn_sync(QcnnCtx* ctx);
int qcnn_ctx_device(const QcnnCtx* ctx);          /* HIP device ordinal of the context */
void* qcnn_ctx_stream(const QcnnCtx* ctx);        /* the hipStream_t the context enqueues on */

/* ---- model ---- */
int qcnn_model_begin(QcnnCtx* ctx, int layer_cnt, const QcnnLayerDesc* layers, int in_c, int in_h, int in_w);
/* Declare the quantisation shape of conv/FC layer `layer` (M sub-spaces, K codewords, Cs dims each).
 * Must precede qcnn_model_commit for every conv/FC layer.  K <= 256 — everything the reference's uint8 assignments can
 * name (include/FileIO.h:128-166); Cs <= 8.  No shipped model has more than 128 code words, and a LUT stage in LDS holds 128
 * rows: a layer with 128 < K <= 256 is cut into ceil(K / 127) pseudo sub-spaces of <= 127 code words + one all-zero row over the
 * same dims (an assignment names its code word in one of them and the zero row in the others: the same sums in the same order)
 * and runs the exact-builder kernels whatever QCNN_OPT_LUT_MODE says — correct, about half the speed of a K <= 128 layer. */
int qcnn_model_set_layer_shape(QcnnCtx* ctx, int layer, int M, int K, int Cs);
/* The reference's PRECISE path (CaffeEva::Init(false): CalcFeatMap_ConvPrec src/CaffeEva.cc:681-758, _FCntPrec :932-966) as
 * an on-device exact baseline: declare conv/FC layer `layer` dense (instead of qcnn_model_set_layer_shape, before commit)
 * and upload bias [Ct] + weights in the reference's FILE layout after commit — conv kernels [Ct][Cin/grp][kh][kw]
 * (convKnl.NN.bin), FC weights [Ct][D] (fcntWei.NN.bin).  Dense and quantised layers may be mixed in one model. */
int qcnn_model_set_layer_dense(QcnnCtx* ctx, int layer);
int qcnn_model_set_layer_weights(QcnnCtx* ctx, int layer, const float* bias, const float* weights_file);
/* Size of the packed parameter arena (biases, permuted codebooks, permuted assignments). */
int qcnn_model_arena_bytes(QcnnCtx* ctx, size_t* bytes);
/* Plan buffers for up to max_batch images.  dev_arena: caller-owned device memory of
 * qcnn_model_arena_bytes() bytes (so that a communicator can broadcast it), or NULL to let the
 * library allocate it. */
int qcnn_model_commit(QcnnCtx* ctx, int max_batch, void* dev_arena);
/* Device address and size of the packed parameter arena (after commit): what a communicator broadcasts. */
int qcnn_model_arena_ptr(QcnnCtx* ctx, void** dev_ptr, size_t* bytes);
/* Checksum of the packed parameter arena as it lies on the device: sum2[0] = sum of its 32-bit words, sum2[1] = a
 * position-weighted sum (both modulo 2^64).  Ranks whose arenas hold the same bytes report the same pair: what a sharded run
 * compares before it trusts a broadcast (qcnn_group_model_broadcast does; bench.py --gpus N does across processes).  Blocking. */
int qcnn_model_arena_checksum(QcnnCtx* ctx, unsigned long long* sum2);
/* Upload one conv/FC layer's parameters from host memory in the reference's FILE layout:
 * bias [Ct]; ctrd [M][K][Cs]; asmt 0-based uint8, [Ct][kh][kw][M] (conv) or [Ct][M] (FC).
 * Performs the PrepCtrdBuf / PrepAsmtBuf permutations into the arena.  After commit. */
int qcnn_model_set_layer_params(QcnnCtx* ctx, int layer, const float* bias, const float* ctrd_file,
                                const uint8_t* asmt_file);
/* Same, with the assignments still bit-packed as the .cbn file holds them (include/FileIO.h:128-166): `cbn_blocks` =
 * the file's payload behind its header (4096-byte blocks of floor(32768 / bits) values, MSB first, 0-based indices in
 * file order [Ct][kh][kw][M] / [Ct][M]), `bits` = its bitCntPerEle.  The packed stream is what crosses PCIe (AlexNet fc6:
 * 5.9 MB instead of 9.4 MB); the decode + PrepAsmtBuf permutation run on the device. */
int qcnn_model_set_layer_params_cbn(QcnnCtx* ctx, int layer, const float* bias, const float* ctrd_file,
                                    const uint8_t* cbn_blocks, size_t cbn_bytes, int bits);
/* Declare every conv/FC layer loaded without uploading: the caller filled the arena itself (e.g. the
 * receiving ranks of an RCCL broadcast of rank 0's arena).  After commit. */
int qcnn_model_mark_loaded(QcnnCtx* ctx);
/* (H, W, C) of feature map l in [0, layer_cnt] */
int qcnn_fm_dims(QcnnCtx* ctx, int l, int* hwc3);

/* ---- forward ---- */
/* Device-resident forward of n <= max_batch images, asynchronous on the context's stream.
 * in_nchw_dev [n][C][H][W] fp32; prob_dev [n][classes] fp32 or NULL; top5_dev [n][5] uint16 or NULL. */
int qcnn_forward(QcnnCtx* ctx, const float* in_nchw_dev, int n, float* prob_dev, uint16_t* top5_dev);
/* Same with the reference's image pre-processing done on the device (BmpImgIO::RmMeanImg + CropImg,
 * src/BmpImgIO.cc:180-224): in_u8_dev [n][C][src_h][src_w] planar 8-bit (B, G, R planes, as
 * BmpImgIO::LoadBmpImg :84-96 stores them), mean_dev [C][src_h][src_w] fp32 or NULL; the network input is
 * the centre crop of float(pixel) - mean.  Bit-identical to qcnn_forward on the host-preprocessed image;
 * a quarter of the host-to-device bytes. */
int qcnn_forward_u8(QcnnCtx* ctx, const uint8_t* in_u8_dev, int src_h, int src_w, const float* mean_dev, int n,
                    float* prob_dev, uint16_t* top5_dev);
/* Blocking convenience: host in, host out (H2D + forward + D2H + sync).  A batch of at least two chunks
 * (QCNN_OPT_HOST_CHUNK) goes through chunk by chunk, uploads overlapped with the previous chunk's layers. */
int qcnn_forward_host(QcnnCtx* ctx, const float* in_nchw_host, int n, float* prob_host, uint16_t* top5_host);
/* Blocking: nb batches one after the other — batch b = n[b] <= max_batch images at in_host[b]; prob_host / top5_host
 * (either may be NULL, and so may single entries) receive the results per batch.  The upload of batch b + 1 runs on a
 * copy stream under the layers of batch b (two device input buffers), results return through pinned buffers: with
 * registered (qcnn_host_register) or pinned input memory the kernels never wait for PCIe. */
int qcnn_forward_host_batches(QcnnCtx* ctx, const float* const* in_host, const int* n, int nb, float* const* prob_host,
                              uint16_t* const* top5_host);
/* Pin / unpin caller memory (hipHostRegister, portable across devices) so that uploads from it are asynchronous DMA
 * transfers.  Errors are reported through qcnn_last_error(NULL). */
int qcnn_host_register(void* ptr, size_t bytes);
int qcnn_host_unregister(void* ptr);
/* Pinned host memory from the HIP runtime (hipHostMalloc, portable): the fastest source / destination of a transfer
 * (measured on MI355X: uploads from it run at about twice the rate of uploads from registered pageable memory). */
int qcnn_host_alloc(size_t bytes, void** out);
int qcnn_host_free(void* ptr);
/* Feature map l of the last forward, images [0, n), NHWC per image, to host (blocking).
 * Requires QCNN_OPT_KEEP_ALL = 1 for maps that the fast path fuses away. */
int qcnn_get_layer_output(QcnnCtx* ctx, int l, int n, float* host_out);
/* Same for images [first, first + n) of the last forward (e.g. one image of the last, ragged panel). */
int qcnn_get_layer_output_range(QcnnCtx* ctx, int l, int first, int n, float* host_out);
/* Run layer `layer` alone on n images: in_host is fm[layer] NHWC per image (FC layers: the flat
 * vector in the order the reference consumes it), out_host receives fm[layer+1] (blocking). */
int qcnn_run_layer(QcnnCtx* ctx, int layer, const float* in_host, int n, float* out_host);

/* How the last launch of conv layer `layer` was cut (QCNN_OPT_SPLIT): *slices = workgroups per split tile (1 = no tile
 * was split), *tiles_unsplit = tiles of the heaviest-first order that ran whole (-1 when nothing was split); sliding
 * kernel (QCNN_OPT_SLIDE): *tiles_unsplit = -2, *slices = segments per output column; a conv or FC layer that ran through
 * its decoded code words (QCNN_OPT_DECODE): *tiles_unsplit = -3, *slices = 1 (FC: slices of the input axis over workgroups);
 * (a conv layer that read the NCHW batch in place: *slices = 2); symmetric workgroups (QCNN_OPT_SYM): *tiles_unsplit = -4;
 * eight-wave symmetric workgroups (QCNN_OPT_SYM8): -5 (conv tile form, *slices = slices per tile under QCNN_OPT_SPLIT, else 1; FC:
 * *slices = splits of the sub-space axis), -6 in their sliding form (*slices = segments per output column;
 * qcnn_get_layer_segments reports the boundaries); their fp16-storage form (QCNN_OPT_LUT_MODE = 2): -7, with fp16 sums too
 * (QCNN_OPT_LUT_MODE = 3): -8. */
int qcnn_get_layer_split(QcnnCtx* ctx, int layer, int* tiles_unsplit, int* slices);
/* Sliding kernel: the row segments [seg_beg9[i], seg_beg9[i + 1]) every output column of the last launch of `layer` was
 * cut into (*n_seg of them; 0 when the layer ran the tile kernel). */
int qcnn_get_layer_segments(QcnnCtx* ctx, int layer, int* seg_beg9, int* n_seg);

/* ---- launch planner (diagnostic; quantized-cnn_amd/csrc/qcnn_planner.h) ---- */
/* The plan and the choice for ONE conv launch, from plain numbers (needs no device, no context):
 *   geom[14] = H, W, Cin, Ho, Wo, Ct, knl, stride, pad, grp, M, Cs, K, panels          opts[8] = QCNN_OPT_SPLIT, _SLIDE, _SYM, _SYM8, _HALF8,
 *   QCNN_OPT_LUT_MODE, flags (1: the layer reads the NCHW input in place, 2: the panels are those of concurrent sub-batches —
 *   QCNN_OPT_STREAMS > 1), partial-sum scratch in Mi floats.
 *   costs[7] (predicted stage-times; 0 = not eligible) = tile kernel, 16-wave sliding, 16-wave symmetric, eight-wave tile, eight-wave
 *   sliding, half-panel tile, half-panel sliding.     choice[13] = family code (what qcnn_get_layer_split reports: -1 tile, -2, -4, -5,
 *   -6, -9, -10), first split tile, slices, segments per column, nine segment boundaries.  Returns non-zero on a malformed geometry. */
int qcnn_plan_conv_query(const int* geom, const int* opts, double* costs, int* choice);

/* ---- timing (QCNN_OPT_PROFILE = 1) ---- */
/* Mean milliseconds of one LAUNCH per layer (a forward issues QCNN_OPT_STREAMS launches per layer, each over
 * its share of the panels) over the forwards recorded since the last reset; ms[layer_cnt]. */
int qcnn_get_layer_ms(QcnnCtx* ctx, float* ms, int* forwards_recorded);
/* Per layer: milliseconds SUMMED over every launch recorded since the last reset, and the number of launches
 * (either output may be NULL); nothing is dropped however many forwards were run. */
int qcnn_get_layer_total_ms(QcnnCtx* ctx, double* total_ms, long long* launches, int* forwards_recorded);
int qcnn_reset_layer_ms(QcnnCtx* ctx);

/* ---- device group: one batch sharded over the GPUs of a node ----
 * One context per device + one RCCL communicator over them (single process).  Images are independent: image i of
 * a batch of n goes to rank i * G / n (contiguous blocks); the only collective is the one-time ncclBroadcast of
 * rank 0's parameter arena over xGMI.  Model calls mirror the per-context ones and apply to every rank. */
typedef struct QcnnGroup QcnnGroup;
/* device_ids == NULL or n_dev <= 0: every visible device.  A device may be listed once; with QCNN_GROUP_ALLOW_DUP=1 in
 * the environment (test rigs with fewer GPUs than ranks) it may repeat: such a group has no RCCL communicator — RCCL
 * refuses two ranks on one device — and "broadcasts" rank 0's arena with device-to-device / peer copies instead; the
 * per-rank contexts, host threads and shard arithmetic are the production ones. */
int qcnn_group_create(const int* device_ids, int n_dev, QcnnGroup** out);
int qcnn_group_destroy(QcnnGroup* grp);
const char* qcnn_group_last_error(const QcnnGroup* grp);   /* grp == NULL: error of qcnn_group_create */
int qcnn_group_size(const QcnnGroup* grp);
QcnnCtx* qcnn_group_ctx(QcnnGroup* grp, int rank);         /* per-rank context: options, dumps, timing */
int qcnn_group_shard_bounds(const QcnnGroup* grp, int n, int rank, int* first, int* count);
int qcnn_group_set_option(QcnnGroup* grp, int option, int value);
int qcnn_group_model_begin(QcnnGroup* grp, int layer_cnt, const QcnnLayerDesc* layers, int in_c, int in_h, int in_w);
int qcnn_group_model_set_layer_shape(QcnnGroup* grp, int layer, int M, int K, int Cs);
int qcnn_group_model_set_layer_dense(QcnnGroup* grp, int layer);
int qcnn_group_model_set_layer_weights(QcnnGroup* grp, int layer, const float* bias, const float* weights_file);   /* rank 0 */
/* max_batch: images of one GLOBAL batch; every rank plans the largest block it can be handed */
int qcnn_group_model_commit(QcnnGroup* grp, int max_batch);
/* uploads to rank 0 only; qcnn_group_model_broadcast then ships the packed arena to the other ranks */
int qcnn_group_model_set_layer_params(QcnnGroup* grp, int layer, const float* bias, const float* ctrd_file,
                                      const uint8_t* asmt_file);
/* Ships rank 0's arena to every other rank (RCCL broadcast over xGMI), then compares a device-side checksum of every rank's
 * arena with rank 0's (qcnn_model_arena_checksum): a mismatch is an error, no rank is declared loaded. */
int qcnn_group_model_broadcast(QcnnGroup* grp, float* elapsed_ms);
/* The checksum pair every rank agreed on at the last broadcast. */
int qcnn_group_arena_checksum(QcnnGroup* grp, unsigned long long* sum2);
/* Blocking: host in, host out; one host thread per GPU runs its block on its own context and stream. */
int qcnn_group_forward_host(QcnnGroup* grp, const float* in_nchw_host, int n, float* prob_host, uint16_t* top5_host);
/* Blocking: nb global batches of n[b] images each, every one sharded over the ranks like qcnn_group_forward_host; a
 * rank's uploads overlap its layers as in qcnn_forward_host_batches. */
int qcnn_group_forward_host_batches(QcnnGroup* grp, const float* const* in_host, const int* n, int nb,
                                    float* const* prob_host, uint16_t* const* top5_host);
/* Asynchronous, device-resident: in_dev[r] / prob_dev[r] / top5_dev[r] are pointers ON RANK r's DEVICE to that rank's block of the
 * batch (qcnn_group_shard_bounds; NULL entries for ranks whose block is empty, prob_dev / top5_dev may be NULL altogether).  Every
 * rank's layers are enqueued on its own stream — no host thread, no PCIe: what a caller that already holds the images on the
 * GPUs uses, and what times the sharded batch without the uploads.  qcnn_group_sync waits for all ranks. */
int qcnn_group_forward(QcnnGroup* grp, const float* const* in_dev, int n, float* const* prob_dev, uint16_t* const* top5_dev);
int qcnn_group_sync(QcnnGroup* grp);

#ifdef __cplusplus
}
#endif
#endif /* QCNN_HIP_H_ */
