// Build-time tuning knobs shared by the streaming kernels (FOR, Dictionary, DZB-fixed, Model).
// Defaults were measured on MI355X; override with -D for experiments (tools/build_variant.sh).
#pragma once

#ifndef GIDDY_STREAM_STEPS_PER_TILE
#define GIDDY_STREAM_STEPS_PER_TILE 4      // 16-byte stores per lane per tile
#endif
#ifndef GIDDY_STREAM_THREADS_PER_BLOCK
#define GIDDY_STREAM_THREADS_PER_BLOCK 256 // workgroup size cap for streaming kernels
#endif
#ifndef GIDDY_FOR_STEPS_PER_TILE
#define GIDDY_FOR_STEPS_PER_TILE GIDDY_STREAM_STEPS_PER_TILE     // Frame of Reference's own tile depth
#endif
#ifndef GIDDY_MODEL_STEPS_PER_TILE
#define GIDDY_MODEL_STEPS_PER_TILE 1       // Model's own tile depth: nothing to load, nothing to overlap
#endif
#ifndef GIDDY_FOR_PATCHED_STAGE_INPUT_UP_TO
#define GIDDY_FOR_PATCHED_STAGE_INPUT_UP_TO 2   // fused FOR: chunks whose rows take at most this many input bytes per lane come in through LDS
#endif
#ifndef GIDDY_FOR_MANY_PATCHED_ROWS_IN_FLIGHT
#define GIDDY_FOR_MANY_PATCHED_ROWS_IN_FLIGHT 4   // fused FOR, ManyPatches build: rows whose inputs are in registers at a time (the other build: 8; here 8 | 4 | 2 at 2 % of patches: 58.5 | 65.3 | 57.8 % of the HBM peak, at 20 %: 48.8 | 57.3 | 55.1)
#endif
#ifndef GIDDY_MODEL_PATCHED_RESIDENT_BLOCKS_CAP
#define GIDDY_MODEL_PATCHED_RESIDENT_BLOCKS_CAP 3   // fused Model: resident workgroups per CU for columns with few patches (0: no throttle).  3 | 5 | 8 at 0.1 % of patches: 72-73 | 69 | 68 % of the HBM peak (2^30 elements), at 3 %: 62.8 | 71.8 | 69.7, at 10 %: 52.7 | 72.2 | 74.3, at 50 %: 31 | 45 | 62 (tools/experiments/model_patched_caps.sh): the wrapper takes 5 from 3 patches per row on (1.2 %: with 3 | 5 at 0.5 % 72 | 65, at 2 % 67 | 72) and none from 16, for callers that state their patch count when resolving
#endif
#ifndef GIDDY_MODEL_PATCHED_ROWS_PER_CHUNK
#define GIDDY_MODEL_PATCHED_ROWS_PER_CHUNK 32  // fused Model: rows of 1 KiB a wavefront produces per seek (8 / 32 / 128 measured: 71-73 / 74-76 / 68-70 % at the best residency)
#endif

// Cache policy of the decoders' 16-byte output stores (memory_access::store_policy, wave64.hpp): 1 = `nt`, 2 = `sc1 nt` (write-through, the line
// is dropped).  Per family, by measurement.
#ifndef GIDDY_DICT_STORE_POLICY
#define GIDDY_DICT_STORE_POLICY 2
#endif
#ifndef GIDDY_FOR_STORE_POLICY
#define GIDDY_FOR_STORE_POLICY 2
#endif
#ifndef GIDDY_MODEL_STORE_POLICY
#define GIDDY_MODEL_STORE_POLICY 2
#endif
#ifndef GIDDY_DELTA_STORE_POLICY
#define GIDDY_DELTA_STORE_POLICY 2
#endif
#ifndef GIDDY_RUN_STAGED_STORE_POLICY
#define GIDDY_RUN_STAGED_STORE_POLICY 2    // run decoders, tiles expanded from LDS-staged values (the look-up path loads between its stores: nt)
#endif
