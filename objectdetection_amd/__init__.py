"""MI355X-native sparse front end of the SECOND detector (voxelize -> VFE -> sparse 3D middle
extractor) behind the reference's spconv / points_to_voxel call signatures.

    objectdetection_amd.spconv            drop-in for the spconv 1.x API the reference uses
    objectdetection_amd.point_cloud_ops   points_to_voxel
    objectdetection_amd.voxel_encoder     VFE classes (reference names / state_dict keys)
    objectdetection_amd.dropin.install()  register them under the reference's import names
    objectdetection_amd.ops               torch-level wrappers over the C ABI (include/lcf.h)

The compute lives in liblcf.so (hand-written gfx950 HIP, objectdetection_amd/csrc); importing this
package does not load it, using any op does and fails loudly if it is missing.
"""
__version__ = "0.1.0"
