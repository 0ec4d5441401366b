/*
 * lcf_cpu.h -- C ABI of liblcf_cpu.so: the GPU-free twin of the voxelizer, for processes that
 * must not touch the GPU.  The reference voxelizes in DataLoader worker processes
 * (second/configs/vkitti/second/base.yaml:120 `num_workers: 3`, second/data/preprocess.py:626-639
 * through second/builder/voxel_builder.py:23-33): those are forked children, where a HIP context
 * cannot be used, so SURVEY.md section 8(b) asks for a fork-safe CPU twin with the signature of the
 * GPU entry point minus workspace and stream.
 *
 * Plain C, no HIP, no threads, no state between calls (fork-safe); every pointer is a HOST
 * pointer; the caller owns all buffers (the call allocates and frees a private hash table with
 * malloc).  Product code: the oracle (oracle/) is not involved, and the GPU path never calls this
 * -- a caller asks for it by name (VoxelGeneratorV2(..., device="cpu"),
 * objectdetection_amd.cpu.points_to_voxel).
 */
#ifndef LCF_CPU_H_
#define LCF_CPU_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Same contract as lcf_voxelize (include/lcf.h), same error codes: points [n,f] f32 row-major;
 * outputs with capacity max_vox rows, every row written (rows >= *voxel_num are zero):
 * voxels [max_vox,max_pts,f] f32, coors [max_vox,coor_stride] i32 ((z,y,x), or (batch_idx,z,y,x)
 * at coor_stride 4), num_pts [max_vox] i32, voxel_num [1] i32.  The sequential first-come loop
 * of the reference (second/utils/simplevis.py:23-50): voxel ids in order of first appearance, the
 * first max_pts points kept per voxel, IEEE fp32 cell arithmetic; cap_mode 0 = `break` when a new
 * voxel would exceed max_vox (numba twin), 1 = `continue` (spconv's C++). */
int lcf_cpu_voxelize(const float *points, int n, int f, const float vs[3], const float range[6],
                     int max_pts, int max_vox, int cap_mode, int batch_idx, int coor_stride,
                     float *voxels, int *coors, int *num_pts, int *voxel_num);

#ifdef __cplusplus
}
#endif
#endif
