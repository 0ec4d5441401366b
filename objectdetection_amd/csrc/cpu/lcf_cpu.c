/* lcf_cpu.c -- liblcf_cpu.so: the GPU-free twin of lcf_voxelize (include/lcf_cpu.h).
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no HIP, no threads, no globals: fork-safe). */
#include "../../../include/lcf_cpu.h"
#include "../../../include/lcf.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CPU_EMPTY 0xFFFFFFFFu

static uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

int lcf_cpu_voxelize(const float *points, int n, int f, const float vs[3], const float range[6],
                     int max_pts, int max_vox, int cap_mode, int batch_idx, int coor_stride,
                     float *voxels, int *coors, int *num_pts, int *voxel_num)
{
    if (n < 0 || f < 3 || max_pts <= 0 || max_vox <= 0 || !vs || !range || !voxels || !coors ||
        !num_pts || !voxel_num || (n > 0 && !points) || cap_mode < 0 || cap_mode > 1 ||
        (coor_stride != 3 && coor_stride != 4))
        return LCF_ERR_ARG;
    int grid[3]; /* x, y, z */
    unsigned long long cells = 1;
    for (int j = 0; j < 3; ++j) {
        /* grid_size = round((hi - lo) / vs) in fp32 (simplevis.py:26-29) */
        volatile float g = (range[3 + j] - range[j]) / vs[j];
        grid[j] = (int)rintf(g);
        if (grid[j] <= 0) return LCF_ERR_ARG;
        cells *= (unsigned long long)grid[j];
    }
    if (cells >= 0xFFFFFFFFull) return LCF_ERR_RANGE;

    memset(voxels, 0, (size_t)max_vox * max_pts * f * sizeof(float));
    memset(coors, 0, (size_t)max_vox * coor_stride * sizeof(int));
    memset(num_pts, 0, (size_t)max_vox * sizeof(int));
    *voxel_num = 0;
    if (n == 0) return LCF_OK;

    /* cell -> voxel id: open addressing over at most min(n, max_vox) cells */
    size_t want = (size_t)(n < max_vox ? n : max_vox) * 2 + 2;
    size_t cap = 16;
    while (cap < want) cap <<= 1;
    uint32_t *keys = (uint32_t *)malloc(cap * sizeof(uint32_t));
    int *ids = (int *)malloc(cap * sizeof(int));
    if (!keys || !ids) {
        free(keys);
        free(ids);
        return LCF_ERR_WORKSPACE;
    }
    memset(keys, 0xFF, cap * sizeof(uint32_t));
    const uint32_t mask = (uint32_t)(cap - 1);

    int k = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = points + (size_t)i * f;
        int c[3];
        int ok = 1;
        for (int j = 0; j < 3; ++j) {
            /* IEEE fp32 subtract and divide, then floor: as numpy does it on float32 arrays */
            volatile float d = p[j] - range[j];
            volatile float q = d / vs[j];
            const float fl = floorf(q);
            if (!(fl >= 0.0f) || fl >= (float)grid[j]) {
                ok = 0;
                break;
            }
            c[j] = (int)fl;
        }
        if (!ok) continue;
        const uint32_t key =
            ((uint32_t)c[2] * (uint32_t)grid[1] + (uint32_t)c[1]) * (uint32_t)grid[0] + (uint32_t)c[0];
        uint32_t h = mix32(key) & mask;
        while (keys[h] != CPU_EMPTY && keys[h] != key) h = (h + 1) & mask;
        int id;
        if (keys[h] == key) {
            id = ids[h];
        } else {
            if (k >= max_vox) {
                if (cap_mode == 0) break; /* numba twin: the loop ends here */
                continue;                 /* spconv C++: this point is dropped */
            }
            id = k++;
            keys[h] = key;
            ids[h] = id;
            int *co = coors + (size_t)id * coor_stride;
            if (coor_stride == 4) *co++ = batch_idx;
            co[0] = c[2];
            co[1] = c[1];
            co[2] = c[0];
        }
        const int m = num_pts[id];
        if (m < max_pts) {
            memcpy(voxels + ((size_t)id * max_pts + m) * f, p, (size_t)f * sizeof(float));
            num_pts[id] = m + 1;
        }
    }
    *voxel_num = k;
    free(keys);
    free(ids);
    return LCF_OK;
}
