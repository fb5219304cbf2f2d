/* The whole forward driven from plain C through include/fsnp.h + the HIP runtime C API: no Python, no torch.
 * Reads a directory written by tests/test_gpu_parity.py::test_forward_from_plain_c (config, named weights, inputs),
 * runs fsnp_create / fsnp_set_weight / fsnp_commit_weights / fsnp_forward and writes the mask back. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fsnp.h"

static void* slurp(const char* dir, const char* name, size_t* bytes) {
    char path[1024];
    FILE* f;
    void* buf;
    snprintf(path, sizeof(path), "%s/%s", dir, name);
    f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(3); }
    fseek(f, 0, SEEK_END);
    *bytes = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    buf = malloc(*bytes ? *bytes : 1);
    if (fread(buf, 1, *bytes, f) != *bytes) { fprintf(stderr, "short read %s\n", path); exit(3); }
    fclose(f);
    return buf;
}

#define CHECK(expr) do { int rc_ = (expr); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, fsnp_last_error()); return 1; } } while (0)
#define HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #expr, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : ".";
    size_t n, wbytes, off = 0;
    fsnp_config* cfg = (fsnp_config*)slurp(dir, "config.bin", &n);
    int32_t* dims = (int32_t*)slurp(dir, "dims.bin", &n);       /* B, F, T, mode */
    unsigned char* wblob = (unsigned char*)slurp(dir, "weights.bin", &wbytes);
    const int B = dims[0], F = dims[1], T = dims[2], mode = dims[3];
    const size_t in_elems = (size_t)B * F * T, out_elems = (size_t)B * 2 * (mode == FSNP_MODE_PARITY ? F / 2 : F) * T;
    float *h_in[3], *d_in[3], *d_out, *h_out;
    const char* names[3] = {"mag.bin", "real.bin", "imag.bin"};
    int64_t strides[3][3];
    fsnp_handle* h = NULL;
    FILE* f;
    char path[1024];
    int i;
    if (n != 4 * sizeof(int32_t)) return 3;
    CHECK(fsnp_create(cfg, &h));
    while (off < wbytes) {                                      /* records: int32 name_len, name, int64 numel, float data */
        int32_t len; int64_t numel; char name[256];
        memcpy(&len, wblob + off, 4); off += 4;
        memcpy(name, wblob + off, (size_t)len); name[len] = 0; off += (size_t)len;
        memcpy(&numel, wblob + off, 8); off += 8;
        CHECK(fsnp_set_weight(h, name, (const float*)(wblob + off), numel));
        off += (size_t)numel * 4;
    }
    CHECK(fsnp_commit_weights(h));
    for (i = 0; i < 3; ++i) {
        h_in[i] = (float*)slurp(dir, names[i], &n);
        if (n != in_elems * 4) return 3;
        HIP(hipMalloc((void**)&d_in[i], n));
        HIP(hipMemcpy(d_in[i], h_in[i], n, hipMemcpyHostToDevice));
        strides[i][0] = (int64_t)F * T; strides[i][1] = T; strides[i][2] = 1;      /* contiguous [B,1,F,T] */
    }
    HIP(hipMalloc((void**)&d_out, out_elems * 4));
    CHECK(fsnp_forward(h, d_in[0], d_in[1], d_in[2], strides, d_out, B, T, mode, 0, B, NULL));
    CHECK(fsnp_check_errors(h));
    h_out = (float*)malloc(out_elems * 4);
    HIP(hipMemcpy(h_out, d_out, out_elems * 4, hipMemcpyDeviceToHost));
    snprintf(path, sizeof(path), "%s/out.bin", dir);
    f = fopen(path, "wb");
    fwrite(h_out, 4, out_elems, f);
    fclose(f);
    printf("ok %d weights, workspace %zu bytes\n", fsnp_num_weights(h), fsnp_workspace_bytes(h, B, T, mode));
    fsnp_destroy(h);
    return 0;
}
