// planner.h - the sub-band planner of libfsnp_hip.so: which kernel runs which sequences.  Host-only code (planner.cpp), driven by a
// cost table; CPU-tested through fsnp_debug_plan_rows / fsnp_debug_plan_rows2 (tests/test_host.py).
#pragma once
#include <vector>

namespace fsnp {

// per-step cost table of the sub-band planner (see default_costs / calibrate_costs); flat layout: costs_to_array (fsnp.h: FSNP_NUM_COSTS)
struct CostTable {
    double ksplit[4][2];       // K-split kernel at 8 / 16 / 32 / 64 units per workgroup x {<= 1, 2} workgroups per CU, launch FULL
    double ksplit1[4];         // the same with ONE row tile (the exchange traffic, hence a step, grows with the tiles in flight)
    double coopn[2][2];        // three-way split, 1 / 2 row tiles per group x {<= 1, 2} workgroups per CU
    double rowtile, rowtile_ex;   // one round of the one-tile-per-CU kernel; relative extra per VALU row
    double rowtile16;             // one round of the half-tile (16-row) kernel (lstm16.hip)
    double hp[2];                 // half-tile ping-pong (lstm_hp.hip, 16 units): ONE row tile, a FULL launch (num_cus / (H / 16) tiles)
    double coopw[3][2];           // wave-owned column split (lstm_coopw.hip) at 32 / 64 / 96 units per workgroup: ONE row tile, a FULL launch
    int calibrated;
};

// ---- plan of the sub-band recurrent model: which kernel runs which sequences.
// The row-tile kernel (lstm.hip) needs >= 256 tiles to fill the chip and costs ~208 us per step however few tiles it
// gets; the column-split kernels pay one inter-workgroup barrier per step instead:
//   <= 42 row tiles  : lstm_coop.hip  (K split, 8..64 hidden units per workgroup, row_tiles * H/units <= CUs)
//   43..170 row tiles: lstm_coopn.hip (3 workgroups x 128 units share 1-2 row tiles)
// A problem is cut into CHUNKS of consecutive sequences that run back to back: e.g. B = 40 (10280 sequences) = one full
// round of the row-tile kernel (8192) + 66 tiles on lstm_coopn.hip instead of two rounds; GRU (column-split only) =
// chunks of <= 170 tiles.  Every chunk owns a slice of the row descriptors / per-row norm tables (slot0) and, if it is
// column-split, of the exchange images and barrier counters (coop_tile0).
struct SbChunk {
    int kind;                  // 0 = row tile, 1 = coop (K split), 2 = coopn, 4 = half tile (lstm16.hip: 16-row tiles, rps = 16),
                               // 7 = runtime-sized (lstm_generic.hip), 8 = half-tile ping-pong (lstm_hp.hip), 9 = wave-owned column split
                               // (lstm_coopw.hip: units = 32 or 64 per workgroup)
    int row0, nrows;           // sequences [row0, row0 + nrows)
    int num_tiles, ex, rps;    // tiles, VALU rows per tile, slots per tile (32 + ex)
    int units, groups, rpg;    // column-split parameters
    int slot0, coop_tile0;
};
struct SbPlan {
    std::vector<SbChunk> chunks;
    int total_slots = 0, coop_tiles = 0;
};

// what the planner needs to know of a handle (fsnp_abi.hip: pctx)
struct PlannerCtx {
    int H = 0, NIN = 0, num_cus = 256, num_cus_real = 256;
    bool gru = false, sb_tcn = false, generic_sb = false, rowtile_ok = true, lstm16_ok = false, hp_ok = false, coopw_ok = false;
    int ih_bf16 = 0, lstm_coop = 1, coop_occ = 1;
    int occ_ksplit[4] = {1, 1, 1, 1}, occ_coopn[2] = {1, 1};
    int coop_hp = 0, coop_w = 0, pipeline = 0;
    bool half_tiles_without_coop = false;   // lstm_coop == 0 plans may still use the (exchange-free) half-tile kernel: fsnp_set_verify's re-run
    double composite_gain = 0.97;
    CostTable cost{};
};

CostTable default_costs();
// the flat table of fsnp_get_costs / fsnp_debug_set_costs (include/fsnp.h: FSNP_NUM_COSTS values)
constexpr int kNumCosts = 27;
void costs_to_array(const CostTable& t, double* out);
void costs_from_array(CostTable& t, const double* in);
// the table a handle starts from: the built-in one scaled to the handle's cell and hidden size (measured at LSTM, H = 384)
CostTable initial_costs(int sb_hidden, bool gru, bool sb_tcn);
int chunk_workgroups(const PlannerCtx& h, const SbChunk& c);
double est_step_us(const PlannerCtx& h, const SbChunk& c);
SbPlan plan_sb(const PlannerCtx& h, int num_rows);       // empty plan = "this device cannot run the model"

}  // namespace fsnp
