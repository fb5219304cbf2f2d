/* Plain-C99 consumer of include/fsnp.h: proves the header is valid C (no C++-isms), that libfsnp_hip.so exports the
 * entry points with C linkage, and that the no-GPU error path is a clean return code + message (never an abort).
 * Built and run by tests/test_host.py::test_header_is_plain_c_and_library_links_from_c. */
#include <stdio.h>
#include <string.h>

#include "fsnp_debug.h"   /* (includes fsnp.h; the host-only planner hook below lives in the debug header) */

int main(void) {
    fsnp_config cfg;
    fsnp_handle* h = NULL;
    int32_t plan[8 * 4];
    int rc, n;
    memset(&cfg, 0, sizeof(cfg));
    cfg.num_freqs = 257; cfg.look_ahead = 2; cfg.sb_num_neighbors = 15; cfg.fb_num_neighbors = 0;
    cfg.tcn_hidden = 512; cfg.num_tcn_blocks = 8; cfg.sb_hidden = 384; cfg.output_size = 2;
    cfg.norm_type = FSNP_NORM_OFFLINE_LAPLACE; cfg.fb_act = FSNP_ACT_RELU; cfg.sb_act = FSNP_ACT_NONE;
    cfg.kersize[0] = 3; cfg.kersize[1] = 5; cfg.kersize[2] = 10;
    cfg.num_groups_in_drop_band = 2; cfg.attention = FSNP_ATT_TSSE;
    cfg.model = FSNP_MODEL_FULLSUBNET_PLUS; cfg.sequence_model = FSNP_SEQ_LSTM;
    printf("version: %s\n", fsnp_version());
    rc = fsnp_create(&cfg, &h);
    printf("create rc=%d handle=%s msg=%s\n", rc, h ? "set" : "null", rc ? fsnp_last_error() : "");
    if (rc == 0) fsnp_destroy(h);                         /* a GPU box: fine too */
    cfg.sb_hidden = 0;                                    /* invalid configuration: must be rejected before any device call */
    rc = fsnp_create(&cfg, &h);
    if (rc == 0) { printf("FAIL: bad sb_hidden accepted\n"); return 1; }
    printf("bad config rc=%d msg=%s\n", rc, fsnp_last_error());
    n = fsnp_debug_plan_rows(8224, 256, 384, 0, 1, 0.97, plan, 4);   /* host-only planner */
    printf("plan chunks=%d first=(%d,%d) second=(%d,%d)\n", n, plan[0], plan[2], plan[8], plan[10]);
    return (n == 2 && plan[2] == 8192 && plan[10] == 32) ? 0 : 2;
}
