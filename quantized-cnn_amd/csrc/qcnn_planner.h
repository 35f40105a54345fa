// qcnn_planner.h — the launch planner of the conv table kernels (host side, no device code): which kernel family runs a conv
// launch of a given geometry and panel count, and how it is cut.
//
// Every family has a cost model in the same unit — "stage-times" = LUT stages of the 16-wave tile kernel (~2500 cycles), tiles /
// strips list-scheduled heaviest first on 256 CUs — calibrated on measurements (LABBOOK.md):
//   tile (k_conv_aprx), whole or with a split tail (qk_conv_plan)                      1 stage = 1 unit
//   16-wave sliding strips (qk_conv_plan_slide)                                         1.03 + 0.3 per source row
//   16-wave symmetric 2x2 x 128 channels (qk_conv_sym_cost)                             1.09
//   eight-wave symmetric tile / sliding form (qk_conv_sym8_cost / _slide_plan)          (2540 + 1.97 x row look-ups per stage) / 2500
//   half-panel eight-wave tile / sliding form (qk_conv_half8_cost / _slide_plan)        (2040 + 0.61 x rows of 64 images) / 2500 per half-panel stage
// qk_plan_conv prices every eligible family once per (layer, launch geometry, options) — the engine caches the result —,
// qk_choose_conv applies the decision rules (a family is taken when forced, or predicted >= 3 % faster than every other one, with
// the measured per-family correction factors).  Pure functions of plain numbers: compiled into libqcnn_hip.so, and by g++ into a
// CPU-side test library (tests/test_planner_cpu.py) — reference: the reference has no counterpart (its layer loop is
// src/CaffeEva.cc:625-670; one CPU thread, nothing to schedule).
#ifndef QCNN_PLANNER_H_
#define QCNN_PLANNER_H_

#include "qcnn_kernels.h"

// what the caller allows / has: the engine's options (QCNN_OPT_*) and which program tables the layer's arena holds
struct QkPlanOptions {
  int split, slide, sym, sym8, half8;     // QCNN_OPT_SPLIT / _SLIDE / _SYM / _SYM8 / _HALF8 as set
  int lutMode;                            // QCNN_OPT_LUT_MODE (1 = f32 MFMA: the only mode the eight-wave families run in)
  int inNchw;                             // the layer reads the network input in place (tile / 16-wave sliding kernels only)
  int concurrent;                         // the forward runs several sub-batches on their own streams (QCNN_OPT_STREAMS > 1): the plan covers the
                                          // panels of all of them, and coarse work items (strips) overlap worse than tiles
  size_t scratchFloats;                   // partial-sum scratch this launch may use (0: no split)
  int hasSlide16, hasSym16, hasSym8, hasSym8Slide, hasHalf8, hasHalf8Slide;   // program tables present
};

// predicted duration of every eligible family, in stage-times (0: not eligible / switched off)
struct QkConvPlan {
  QkSplitPlan plan;                       // tile kernel, whole or with a split tail
  double slideCost; int segN, segBeg[9];  // 16-wave sliding strips + their segments
  double symCost;                         // 16-wave symmetric
  double sym8Cost; int sym8Z;             // eight-wave tile form (every tile in sym8Z slices)
  double sym8sCost; int seg8N, seg8Beg[9];
  double half8Cost;                       // half-panel eight-wave tile form
  double half8sCost; int segHN, segHBeg[9];
};

// kernel family codes = what qcnn_get_layer_split reports as *tiles_unsplit for the conv table kernels
enum QkConvFamily { QK_FAM_TILE = -1, QK_FAM_SLIDE16 = -2, QK_FAM_SYM16 = -4, QK_FAM_SYM8 = -5, QK_FAM_SYM8_SLIDE = -6,
                    QK_FAM_HALF8 = -9, QK_FAM_HALF8_SLIDE = -10 };
struct QkConvChoice {
  int family;                             // QkConvFamily
  int splitFrom, Z;                       // QK_FAM_TILE: tiles from rank splitFrom on in Z slices (Z <= 1: whole); QK_FAM_SYM8: Z slices per tile
  int nSeg, segBeg[9];                    // sliding families: segments per column
};

constexpr double QK_SYM8_STAGE_FACTOR = 0.97;   // scale of the eight-wave stage prices (their list schedule over-prices the last round by ~3 %)
constexpr double QK_SLIDE8_FACTOR = 1.15;       // a planner unit of the 16-wave / eight-wave sliding forms against one of the tile forms (measured 1.04 - 1.11 us against 0.89 - 0.93)
constexpr double QK_CONCURRENT_STRIP_FACTOR = 1.08;   // sliding forms when sub-batches run concurrently (measured: AlexNet conv5 on two streams, sliding 9.45 against tile form 9.36 ms per step)
constexpr double QK_HALF8_SLIDE_FACTOR = 1.25;  // ... of the half-panel sliding form against the half-panel tile form (measured 1.21 - 1.28 against 1.01 - 1.07)

QkConvPlan qk_plan_conv(const ConvParams& p, const QkPlanOptions& o);
QkConvChoice qk_choose_conv(const QkConvPlan& pl, const QkPlanOptions& o);

#endif  // QCNN_PLANNER_H_
