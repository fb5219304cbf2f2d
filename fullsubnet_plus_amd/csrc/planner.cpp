// planner.cpp - the sub-band planner (see planner.h): cost table, launch shapes, shortest path over tile counts.  Host-only.
#include <algorithm>
#include <utility>
#include <vector>

#include "fsnp_common.h"
#include "planner.h"

namespace fsnp {

// Per-step cost (microseconds) of every kernel shape the planner can choose.  The defaults are the round-1 measurements
// (profiles/r01_column_split.md); fsnp_forward replaces them once per process and device with values MEASURED on the
// device (calibrate_costs below: each shape timed at two step counts, cost = slope), so a kernel change can no longer
// silently mis-plan.  [.][1] = the launch has more workgroups than CUs, i.e. two are co-resident per CU and share its
// matrix pipes (their hand-off stalls then overlap: two independent row tiles per CU) - only planned when the kernels'
// occupancy allows it (coop_occ >= 2) and, without a calibration, priced so that it is never chosen.
CostTable default_costs() {
    CostTable t{};
    // round-3 measurements (fsnp_measure_costs on an MI355X, profiles/r03_planner_costs.json - after the arrival counters got a
    // 128-byte line each, which took 35 % off a full K-split launch at 8 units and 14 % at 16): K split (serial schedule at 8
    // units, layer-skewed from 16 up): a full launch / one row tile; three-way split at 85 / 170 row tiles
    const double ks[4] = {8.4, 14.2, 24.8, 49.2}, k1[4] = {8.0, 13.3, 22.7, 45.6}, cn[2] = {84.0, 154.0};
    for (int i = 0; i < 4; ++i) { t.ksplit[i][0] = ks[i]; t.ksplit[i][1] = 3.0 * ks[i]; t.ksplit1[i] = k1[i]; }
    for (int i = 0; i < 2; ++i) { t.coopn[i][0] = cn[i]; t.coopn[i][1] = 2.2 * cn[i]; }
    t.rowtile = 206.0; t.rowtile_ex = 0.11; t.rowtile16 = 103.0;
    // half-tile ping-pong launches: one row tile / a full launch of 10.  Round 6: they run on the wave-owned variant (lstm_hpw.hip):
    // 10.9 ... 11.7 us per step in 128-step runs (profiles/r06_hpw_times.txt), 9.6 / 10.1 in fsnp_measure_costs' short runs on zeros;
    // lstm_hp.hip (FSNP_HP_WAVE=0; H = 384 with more than 40 input features): 12.3 / 12.9
    t.hp[0] = 10.4; t.hp[1] = 11.0;
    // round 5: wave-owned column split (lstm_coopw.hip) at 32 / 64 units per workgroup: one row tile / a full launch of 21 / 42
    // (profiles/r05_column_split.md; a 96-unit instantiation - 64 tiles x 4 workgroups - measured 63.3 us: no better than a 42-tile
    // launch at 64 units + a 21-tile launch at 32, and it spilled registers: not built)
    // round 6: 96 units per workgroup (NT = 3: 4 workgroups per row tile, 64 row tiles per launch - B = 8's 65 tiles as 64 + 1 instead of
    // 42 + 21 + 2).  With the transposed product the cells come straight out of the accumulators, the staging tile is gone and the
    // instantiation fits without spills (406 registers); 4 loads per 12 MFMAs in its k-loops (NT = 1: 2 per 4)
    const double cw1[3] = {21.0, 37.0, 52.5}, cwf[3] = {22.3, 38.5, 56.5};      // (profiles/r06_coopw_times.txt)
    for (int i = 0; i < 3; ++i) { t.coopw[i][0] = cw1[i]; t.coopw[i][1] = cwf[i]; }
    return t;
}
// flat layout (include/fsnp.h, fsnp_get_costs): [0..7] K split full launch x {one, two per CU} at 8 / 16 / 32 / 64 units, [8..11] three-way
// split 1 / 2 row tiles per group x {one, two per CU}, [12] one tile per CU, [13] relative extra per VALU row, [14..17] K split with ONE row
// tile, [18] half tile per CU, [19..20] half-tile ping-pong one tile / full launch, [21..22] wave-owned split at 32 / 64 units per
// workgroup, a full launch, [23..24] the same with ONE row tile, [25..26] (round 6) the 96-unit instantiation: a full launch, one row tile
// (kNumCosts = 27; appended, so that a 25-value table of the round-5 layout is still a valid prefix)
void costs_to_array(const CostTable& t, double* out) {
    for (int i = 0; i < 4; ++i) { out[2 * i] = t.ksplit[i][0]; out[2 * i + 1] = t.ksplit[i][1]; out[14 + i] = t.ksplit1[i]; }
    for (int i = 0; i < 2; ++i) { out[8 + 2 * i] = t.coopn[i][0]; out[9 + 2 * i] = t.coopn[i][1]; }
    out[12] = t.rowtile; out[13] = t.rowtile_ex; out[18] = t.rowtile16;
    out[19] = t.hp[0]; out[20] = t.hp[1];
    for (int i = 0; i < 2; ++i) { out[21 + i] = t.coopw[i][1]; out[23 + i] = t.coopw[i][0]; }
    out[25] = t.coopw[2][1]; out[26] = t.coopw[2][0];
}
void costs_from_array(CostTable& t, const double* in) {
    for (int i = 0; i < 4; ++i) { t.ksplit[i][0] = in[2 * i]; t.ksplit[i][1] = in[2 * i + 1]; t.ksplit1[i] = in[14 + i]; }
    for (int i = 0; i < 2; ++i) { t.coopn[i][0] = in[8 + 2 * i]; t.coopn[i][1] = in[9 + 2 * i]; }
    t.rowtile = in[12]; t.rowtile_ex = in[13]; t.rowtile16 = in[18];
    t.hp[0] = in[19]; t.hp[1] = in[20];
    for (int i = 0; i < 2; ++i) { t.coopw[i][1] = in[21 + i]; t.coopw[i][0] = in[23 + i]; }
    t.coopw[2][1] = in[25]; t.coopw[2][0] = in[26];
}
// the table a handle starts from: the built-in one scaled to the handle's cell and hidden size (measured at LSTM, H = 384)
CostTable initial_costs(int sb_hidden, bool gru, bool sb_tcn) {
    CostTable t = default_costs();
    if (gru) t.rowtile *= 0.75;   // three of the four gate tiles per k-group
    if (sb_hidden != 384 && !sb_tcn) {     // scale by the work per step
        const double f = sb_hidden / 384.0;
        for (int i = 0; i < 4; ++i) { t.ksplit[i][0] *= f; t.ksplit[i][1] *= f; t.ksplit1[i] *= f; }
        t.hp[0] *= f; t.hp[1] *= f;
        for (int i = 0; i < 3; ++i) { t.coopw[i][0] *= f; t.coopw[i][1] *= f; }
        for (int i = 0; i < 2; ++i) { t.coopn[i][0] *= f; t.coopn[i][1] *= f; }
        t.rowtile *= f * f;
    }
    return t;
}
static int units_index(int units) { return units <= 8 ? 0 : units <= 16 ? 1 : units <= 32 ? 2 : 3; }
int chunk_workgroups(const PlannerCtx& h, const SbChunk& c) {
    if (c.kind == 1) return c.num_tiles * (h.H / c.units);
    if (c.kind == 2) return c.groups * (h.H / 128);
    if (c.kind == 8) return c.num_tiles * (h.H / 16);
    if (c.kind == 9) return c.num_tiles * (h.H / c.units);
    return c.num_tiles;
}
double est_step_us(const PlannerCtx& h, const SbChunk& c) {
    const int dbl = chunk_workgroups(h, c) > h.num_cus_real ? 1 : 0;
    if (c.kind == 1) {
        const int ui = units_index(c.units);
        const int cap = h.num_cus_real / (h.H / c.units);
        if (dbl || cap <= 1) return h.cost.ksplit[ui][dbl];
        const double f = (double)(c.num_tiles - 1) / (cap - 1);              // 1 tile .. a full launch
        return h.cost.ksplit1[ui] + (h.cost.ksplit[ui][0] - h.cost.ksplit1[ui]) * (f < 1.0 ? f : 1.0);
    }
    if (c.kind == 2) return h.cost.coopn[c.rpg == 1 ? 0 : 1][dbl];
    if (c.kind == 8) {
        const int cap = h.num_cus_real / (h.H / 16);
        if (cap <= 1) return h.cost.hp[0];
        const double f = (double)(c.num_tiles - 1) / (cap - 1);
        return h.cost.hp[0] + (h.cost.hp[1] - h.cost.hp[0]) * (f < 1.0 ? f : 1.0);
    }
    if (c.kind == 9) {
        const int wi = c.units / 32 - 1, cap = h.num_cus_real / (h.H / c.units);
        if (cap <= 1) return h.cost.coopw[wi][0];
        const double f = (double)(c.num_tiles - 1) / (cap - 1);
        return h.cost.coopw[wi][0] + (h.cost.coopw[wi][1] - h.cost.coopw[wi][0]) * (f < 1.0 ? f : 1.0);
    }
    // (bf16 ih-GEMM mode: the half-tile kernel streams 12 bf16 k-steps instead of 48 fp32 k-groups for layer 1's ih product - measured
    //  11.42 vs 14.16 ms at 4096 sequences, profiles/r04_bf16_half_tile_bench.jsonl: the fp32 entry scaled by that ratio)
    if (c.kind == 4) return cdiv(c.num_tiles, h.num_cus) * h.cost.rowtile16 * (h.ih_bf16 == 1 ? 0.81 : 1.0);
    // (the same for the one-tile-per-CU LSTM kernel: 155 us per round against 206 in fp32 - BASELINE configs[4], 20.9 vs 27.5 ms at
    //  B = 32, profiles/r04_bench_configs.md.  Without this the cheaper column-split kernels of round 5 made the planner leave the
    //  bf16 kernel for a half-tile round + three wave-owned launches at the fp32 kernel's price: 27.6 ms, profiles/r05_bench_configs.md)
    const double bf = (h.ih_bf16 == 1 && !h.gru) ? 0.755 : 1.0;
    return cdiv(c.num_tiles, h.num_cus) * h.cost.rowtile * bf * (1.0 + h.cost.rowtile_ex * c.ex);
}
static SbChunk rowtile_chunk(const PlannerCtx& h, int row0, int nrows) {
    if (h.gru || h.H != 384) return SbChunk{0, row0, nrows, cdiv(nrows, 32), 0, 32, 0, 0, 0, 0, 0};   // VALU rows: LSTM at H = 384 only
    const LstmPlan lp = plan_lstm_tiles(nrows, h.num_cus);
    return SbChunk{0, row0, nrows, lp.num_tiles, lp.ex, lp.rows_per_slot_tile, 0, 0, 0, 0, 0};
}
// Column-split launches for `nrows` sequences (any count): the cheapest sequence of launches by the cost table.  A launch
// costs the same per step whether its kernel is full or not, so this is a shortest path over tile counts: best[t] = min over
// launch shapes c (K split at 8..64 units, one or two row tiles per three-workgroup group; one or - if the kernels fit -
// two workgroups per CU) of cost(c) + best[t - min(t, capacity(c))].  E.g. with the round-1 table: 128 tiles = 85 one per
// group (76 us) + 42 K-split (55) + 1 (9) instead of two per group (151); 97 = 85 + 12 (76 + 29).
static std::vector<SbChunk> plan_columns(const PlannerCtx& h, int row0, int nrows) {
    std::vector<SbChunk> out;
    const int S3 = h.H / 128;
    if (h.H < 128 || h.num_cus_real / S3 <= 0) return out;   // fewer CUs than one group needs: no column-split plan
    const int T = cdiv(nrows, 32);
    struct Shape { int kind, units, rpg, cap, dbl; };
    std::vector<Shape> shapes;
    for (int occ = 1; occ <= (h.coop_occ >= 2 ? 2 : 1); ++occ) {          // two per CU: only shapes whose kernel fits twice
        const int slots = h.num_cus_real * occ;
        for (int u = 8; u <= 64; u *= 2)
            if (h.H % u == 0 && slots / (h.H / u) > 0 && h.occ_ksplit[units_index(u)] >= occ)
                shapes.push_back({1, u, 0, slots / (h.H / u), occ - 1});
        for (int rpg = 1; rpg <= 2; ++rpg)
            if (h.occ_coopn[rpg - 1] >= occ) shapes.push_back({2, 0, rpg, (slots / S3) * rpg, occ - 1});
        // half-tile ping-pong (lstm_hp.hip): H / 16 workgroups per row tile
        if (occ == 1 && h.hp_ok && h.coop_hp && slots / (h.H / 16) > 0) shapes.push_back({8, 16, 0, slots / (h.H / 16), 0});
        // wave-owned column split (lstm_coopw.hip): H / (32 NT) workgroups per row tile, one per CU
        for (int nt = 1; nt <= 3 && occ == 1 && h.coopw_ok && h.coop_w; ++nt)
            if (h.H % (32 * nt) == 0 && slots / (h.H / (32 * nt)) > 0) shapes.push_back({9, 32 * nt, 0, slots / (h.H / (32 * nt)), 0});
    }
    auto shape_cost = [&](const Shape& sh, int n) {             // n tiles on this shape (n <= cap)
        SbChunk c{sh.kind, 0, n * 32, n, 0, 32, sh.units, sh.kind == 2 ? cdiv(n, sh.rpg) : 0, sh.rpg, 0, 0};
        return est_step_us(h, c) + 1.2;                         // + a launch (prologue / drain, amortised over ~100 steps): fewer chunks win near-ties
    };
    std::vector<double> best(T + 1, 0.0);
    std::vector<int> pick(T + 1, -1);
    for (int t = 1; t <= T; ++t) {
        best[t] = 1e30;
        for (int i = 0; i < (int)shapes.size(); ++i) {
            const int n = t < shapes[i].cap ? t : shapes[i].cap;
            if (shapes[i].kind == 2 && shapes[i].rpg == 2 && n < 2) continue;
            const double c = shape_cost(shapes[i], n) + best[t - n];
            if (c < best[t] - 1e-9) { best[t] = c; pick[t] = i; }
        }
        if (pick[t] < 0) return out;
    }
    std::vector<std::pair<int, int>> taken;                     // (tiles, shape), largest first
    for (int t = T; t > 0;) {
        const Shape& sh = shapes[pick[t]];
        const int n = t < sh.cap ? t : sh.cap;
        taken.push_back({n, pick[t]});
        t -= n;
    }
    std::stable_sort(taken.begin(), taken.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
    int r0 = row0, left = nrows;
    for (const auto& tk : taken) {
        const Shape& sh = shapes[tk.second];
        const int rows = tk.first * 32 < left ? tk.first * 32 : left;
        SbChunk c{sh.kind, r0, rows, tk.first, 0, 32, sh.units, sh.kind == 2 ? cdiv(tk.first, sh.rpg) : 0, sh.rpg, 0, 0};
        out.push_back(c);
        r0 += rows; left -= rows;
    }
    return out;
}

SbPlan plan_sb(const PlannerCtx& h, int num_rows) {
    SbPlan p;
    auto push = [&](SbChunk c) {
        c.slot0 = p.total_slots; p.total_slots += c.num_tiles * c.rps;
        c.coop_tile0 = p.coop_tiles; if (c.kind == 1 || c.kind == 2 || c.kind == 8 || c.kind == 9) p.coop_tiles += c.num_tiles;
        p.chunks.push_back(c);
    };
    if (h.sb_tcn) { push(SbChunk{0, 0, num_rows, cdiv(num_rows, 32), 0, 32, 0, 0, 0, 0, 0}); return p; }   // no recurrent kernel
    if (h.generic_sb) {                                        // the runtime-sized kernel: workgroups of rg sequences, one launch
        const int rg = lstm_generic_rows_per_group(h.H, h.NIN, num_rows, h.num_cus_real);
        if (rg > 0) push(SbChunk{7, 0, num_rows, cdiv(num_rows, rg), 0, rg, 0, 0, rg, 0, 0});
        return p;
    }
    // (+ 1.2 us per launch, as inside plan_columns: prologue / drain amortised over ~100 steps - fewer launches win near-ties)
    auto cost_of = [&](const std::vector<SbChunk>& v) { double c = 0; for (const SbChunk& k : v) c += est_step_us(h, k) + 1.2; return c; };
    const bool rowtile_ok = h.rowtile_ok;                     // a one-tile-per-CU kernel exists for this cell / size
    const bool coop_on = h.lstm_coop != 0 || !rowtile_ok;     // (without one the column-split kernels are the only path)
    const SbChunk whole = rowtile_chunk(h, 0, num_rows);
    // bf16-ih mode (configs[4]) only changes the row-tile kernel: sequences that run on a column-split kernel (small
    // batches, remainder tiles) stay fp32 - more accurate and, there, faster
    if (!coop_on) {
        // (the exchange-free re-run of fsnp_set_verify: up to one round of 16-row half tiles costs half a round of 32-row tiles; plain
        //  FSNP_LSTM_COOP=0 handles keep the row-tile kernel, whose plans their users and tests know)
        const SbChunk all16{4, 0, num_rows, cdiv(num_rows, 16), 0, 16, 0, 0, 0, 0, 0};
        if (h.half_tiles_without_coop && h.lstm16_ok && est_step_us(h, all16) < est_step_us(h, whole)) push(all16);
        else push(whole);
        return p;
    }
    // candidates: everything column-split; one launch of the row-tile kernel (VALU rows / extra rounds as needed); full
    // rounds of the row-tile kernel + the remainder column-split (must be `composite_gain` cheaper than the single launch)
    const int full = h.num_cus * 32, q = num_rows / full, rem = num_rows - q * full;
    std::vector<SbChunk> best;
    double best_cost = 1e30;
    if (cdiv(num_rows, 32) <= 4 * h.num_cus_real || !rowtile_ok) {        // (bounded: the shortest path is O(tiles x shapes))
        const std::vector<SbChunk> cols = plan_columns(h, 0, num_rows);
        if (!cols.empty()) { best = cols; best_cost = cost_of(cols); }
    }
    if (rowtile_ok) {
        const double cw = est_step_us(h, whole) + 1.2;
        if (cw < best_cost) { best = {whole}; best_cost = cw; }
        if (q >= 1 && rem > 0) {
            std::vector<SbChunk> comp{SbChunk{0, 0, q * full, q * h.num_cus, 0, 32, 0, 0, 0, 0, 0}};
            const std::vector<SbChunk> rc = plan_columns(h, q * full, rem);
            if (!rc.empty()) {
                comp.insert(comp.end(), rc.begin(), rc.end());
                const double cc = cost_of(comp);
                if (cc < h.composite_gain * cw && cc < best_cost) { best = comp; best_cost = cc; }
            }
        }
    }
    // half tiles (lstm16.hip, LSTM at the default sizes, fp32): 16 CUs-worth of sequences per round in about half the time of a
    // 32-row round - the cheapest shape between the column-split kernels' range and a chip-filling round (parity-mode B = 32:
    // 4096 sequences = 256 half tiles, one launch), alone or as one full round + a column-split remainder
    if (h.lstm16_ok) {                           // (fp32 and, round 4, the bf16 ih-GEMM mode)
        const int per_round = h.num_cus * 16;
        const SbChunk all16{4, 0, num_rows, cdiv(num_rows, 16), 0, 16, 0, 0, 0, 0, 0};
        const double c_all = est_step_us(h, all16) + 1.2;
        if (c_all < best_cost) { best = {all16}; best_cost = c_all; }
        if (num_rows > per_round) {
            std::vector<SbChunk> comp{SbChunk{4, 0, per_round, h.num_cus, 0, 16, 0, 0, 0, 0, 0}};
            const std::vector<SbChunk> rc = plan_columns(h, per_round, num_rows - per_round);
            if (!rc.empty()) {
                comp.insert(comp.end(), rc.begin(), rc.end());
                const double cc = cost_of(comp);
                // (a composite of a half-tile round + column-split launches obeys composite_gain like the row-tile composite above:
                //  with the 96-unit wave-owned launches of round 6 such a plan came within 2 % of ONE row-tile launch at B = 32)
                const double whole_cost = rowtile_ok ? h.composite_gain * (est_step_us(h, whole) + 1.2) : 1e30;
                if (cc < best_cost && cc < whole_cost) { best = comp; best_cost = cc; }
            }
        }
    }
    for (const SbChunk& c : best) push(c);                     // empty = "this device cannot run the model"
    return p;
}

}  // namespace fsnp
