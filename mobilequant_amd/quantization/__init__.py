from .fp_ops import ElementwiseAdd, ElementwiseMul, FMatMul, HFRMSNorm, L2Norm  # noqa: F401
from .qmodule import (QGELU, QLayerNorm, QLinear, QMatMul, QRMSNorm, QSiLU, QuantConfig, QuantLinear,  # noqa: F401
                      Quantizer, compute_min_max_from_scale_offset, compute_min_max_from_tensor,
                      compute_scale_offset_from_min_max, create_fp_model, create_sim_qmodel,
                      create_weight_only_qmodel, export_act_range, export_qcfg, round_ste, set_scale_and_offset,
                      update_qcfg, wire_integer_inputs, fuse_gated_mlp, int8_coverage)
from .smooth import (smooth_fc_fc_inplace, smooth_fc_fc_temporary, smooth_lm_inplace, smooth_lm_temporary,  # noqa: F401,E402
                     smooth_ln_fcs_inplace, smooth_ln_fcs_temporary, smooth_q_k_inplace, smooth_q_k_temporary, truncate_number)
