"""moda_amd -- MI355X (gfx950) implementation of MoDA's per-ray rendering hot path.

Mirrors the reference's call surface (nnutils/rendering.py, nerf.py, dual_quat.py and the skinning
subset of geom_utils.py); all arithmetic runs in libmoda_hip.so (include/moda_hip.h).
"""
from .nerf import Embedding, NeRF, NeRFUnc, set_precision, get_precision  # noqa: F401
from .rendering import render_rays, inference, inference_deform, sample_pdf  # noqa: F401
from .geom_utils import (evaluate_mlp, bone_transform, vec_to_sim3, gauss_mlp_skinning, mlp_skinning,  # noqa: F401
                         skinning, neu_dbs, dqs_blend_skinning)
from .dual_quat import (q_normalize, q_mul, dq_mul, dq_normalize, dq_quaternion_conjugate,  # noqa: F401
                        dq_combined_conjugate, dq_inverse)
from .loss_utils import (visibility_loss, compute_pts_exp, feat_match_loss, feat_match, kp_reproj_loss,  # noqa: F401
                         kp_reproj, eikonal_loss, nerf_gradient, compute_gradients_sdf)
from .feeders import (raycast, sample_xy, chunk_rays, FrameCode, DQ_RTHead, correct_bones, correct_rest_pose,  # noqa: F401
                      update_rays, update_delta_rts)
from .mesh_queries import warp_bw, warp_fw, query_volume  # noqa: F401
from . import checkpoint  # noqa: F401
from . import overflow  # noqa: F401
from .autograd import set_train_precision, get_train_precision, GradBucket  # noqa: F401
