/* A non-Python host of libf8net.so: plain C99 against include/f8net.h.
 *
 *   gcc -std=c99 -Wall -Iinclude examples/host_demo.c -Lf8net_amd -lf8net -Wl,-rpath,$PWD/f8net_amd -o build/host_demo
 *
 * Builds a two-layer integer net (3x3 conv + ReLU -> 1x1 conv, residual join with the first conv's output) through the
 * builder API, finalizes it (planning needs no GPU) and prints the plan.  With a GPU (argv[1] == "run") it also uploads
 * and runs it on zeros; device buffers come from the HIP runtime, which a real host links anyway. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "f8net.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s failed: %s\n", #x, f8_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    enum { C = 32, H = 8, W = 8, N = 2 };
    static int32_t w1[C * C * 9], w2[C * C], b1[C], b2[C];
    for (int i = 0; i < C * C * 9; ++i) w1[i] = (i * 7) % 11 - 5;
    for (int i = 0; i < C * C; ++i) w2[i] = (i * 5) % 9 - 4;
    for (int i = 0; i < C; ++i) { b1[i] = 100 * i; b2[i] = -50 * i; }

    printf("libf8net version %d, %d device(s)\n", f8_version(), f8_device_count());
    f8_net* net = f8_net_create();
    if (!net) return 1;
    int x = f8_net_input(net, C, H, W, /*fraclen*/ 8);
    CHECK(x);
    f8_conv_desc d1; memset(&d1, 0, sizeof d1);
    d1.cin = C; d1.cout = C; d1.kernel = 3; d1.stride = 1; d1.pad = 1; d1.groups = 1;
    d1.weight_fl = 5; d1.input_fl = 6; d1.input_signed = 0; d1.quant_input = 1; d1.relu = 1;
    int t1 = f8_net_conv(net, x, &d1, w1, b1);
    CHECK(t1);
    f8_conv_desc d2 = d1;
    d2.kernel = 1; d2.pad = 0; d2.weight_fl = 6; d2.input_fl = 5; d2.relu = 0;
    int t2 = f8_net_conv(net, t1, &d2, w2, b2);
    CHECK(t2);
    int t3 = f8_net_add(net, t2, t1, /*relu*/ 1);
    CHECK(t3);
    CHECK(f8_net_output(net, t3, /*as_float*/ 0));
    CHECK(f8_net_finalize(net, N));
    size_t need = f8_net_describe(net, NULL, 0);
    char* plan = (char*)malloc(need);
    f8_net_describe(net, plan, need);
    printf("%s", plan);
    free(plan);
    printf("output fraclen %d, %zu elements per image, %d launches\n", f8_net_output_fraclen(net), f8_net_output_elems(net),
           f8_net_num_launches(net));
    if (argc > 1 && strcmp(argv[1], "run") == 0) {
        CHECK(f8_net_upload(net));
        printf("uploaded: arena %zu B, weights %zu B\n", f8_net_arena_bytes(net), f8_net_weight_bytes(net));
    }
    f8_net_destroy(net);
    return 0;
}
