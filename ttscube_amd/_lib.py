"""ctypes binding of libttscube_hip.so (the C ABI declared in include/ttscube_hip.h).

The product path fails loudly when the HIP extension is missing — there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libttscube_hip.so')


class TTSCError(RuntimeError):
    pass


class Conv1dCfg(C.Structure):
    _fields_ = [('in_channels', C.c_int32), ('out_channels', C.c_int32), ('kernel_size', C.c_int32),
                ('stride', C.c_int32), ('padding', C.c_int32), ('dilation', C.c_int32), ('transposed', C.c_int32), ('groups', C.c_int32)]


class Conv1dEpilogue(C.Structure):
    _fields_ = [('in_scale', C.c_float), ('in_slope', C.c_float), ('out_scale', C.c_float),
                ('out_act', C.c_int32), ('accumulate', C.c_int32), ('gate_dev', C.c_void_p), ('gate_slope', C.c_float)]


MAX_UPS, MAX_RB, MAX_DIL = 8, 8, 8


class WBankEntry(C.Structure):
    """ttsc_wbank_entry (include/ttscube_hip.h)"""
    _fields_ = [('v', C.c_void_p), ('g', C.c_void_p), ('w', C.c_void_p), ('norm', C.c_void_p), ('amax', C.c_void_p), ('pack_fwd', C.c_void_p),
                ('pack_dgrad', C.c_void_p), ('Cin', C.c_int32), ('Cout', C.c_int32), ('K', C.c_int32), ('groups', C.c_int32), ('stride', C.c_int32)]


class HifiganCfg(C.Structure):
    _fields_ = [('num_mels', C.c_int32), ('upsample_initial_channel', C.c_int32), ('resblock', C.c_int32),
                ('num_upsamples', C.c_int32), ('upsample_rates', C.c_int32 * MAX_UPS),
                ('upsample_kernel_sizes', C.c_int32 * MAX_UPS), ('num_kernels', C.c_int32),
                ('resblock_kernel_sizes', C.c_int32 * MAX_RB), ('num_dilations', C.c_int32 * MAX_RB),
                ('resblock_dilation_sizes', (C.c_int32 * MAX_DIL) * MAX_RB)]


ACT_NONE, ACT_TANH, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3
PREC_FP32, PREC_F16X3 = 0, 1


class WavernnCfg(C.Structure):
    _fields_ = [('H', C.c_int32), ('num_layers', C.c_int32), ('use_lowres', C.c_int32), ('upsample', C.c_int32),
                ('upsample_low', C.c_int32), ('S', C.c_int32), ('n_mel', C.c_int32), ('out_kind', C.c_int32)]


WR_OUT_MULAW, WR_OUT_RAW, WR_OUT_MOL, WR_OUT_GM, WR_OUT_BETA = 0, 1, 2, 3, 4
WR_MODE_ARGMAX, WR_MODE_NOISE, WR_MODE_PHILOX = 0, 1, 2

_lib = None

# symbol -> (restype, argtypes); also the list the CPU test checks against include/ttscube_hip.h
SIGNATURES = {
    'ttsc_version': (C.c_char_p, []),
    'ttsc_last_error': (C.c_char_p, []),
    'ttsc_device_count': (C.c_int, []),
    'ttsc_conv1d_create': (C.c_int, [C.POINTER(Conv1dCfg), C.POINTER(C.c_void_p)]),
    'ttsc_conv1d_set_weight': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttsc_conv1d_set_precision': (C.c_int, [C.c_void_p, C.c_int32]),
    'ttsc_conv1d_out_len': (C.c_int64, [C.c_void_p, C.c_int64]),
    'ttsc_conv1d_set_activation_scale': (C.c_int, [C.c_void_p, C.c_float]),
    'ttsc_conv1d_get_activation_scale': (C.c_float, [C.c_void_p]),
    'ttsc_absmax': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'ttsc_hifigan_calibrate': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ttsc_hifigan_get_activation_scale': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_float)]),
    'ttsc_hifigan_set_activation_scales': (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int32]),
    'ttsc_hifigan_recalibrations': (C.c_int32, [C.c_void_p]),
    'ttsc_hifigan_set_branch_streams': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttsc_hifigan_set_range_check': (C.c_int, [C.c_void_p, C.c_int32]),
    'ttsc_hifigan_range_status': (C.c_int32, [C.c_void_p, C.c_void_p]),
    'ttsc_conv1d_set_nonfinite_flag': (C.c_int, [C.c_void_p, C.c_void_p]),
    'ttsc_conv1d_in_channels': (C.c_int32, [C.c_void_p]),
    'ttsc_conv1d_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.POINTER(Conv1dEpilogue), C.c_void_p]),
    'ttsc_conv1d_forward_ragged': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                             C.POINTER(Conv1dEpilogue), C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttsc_conv1d_forward_pitched': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                              C.POINTER(Conv1dEpilogue), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'ttsc_rbchain_supported': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32]),
    'ttsc_rbchain_forward': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int32, C.c_int64,
                                       C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    'ttsc_rbchain_post_supported': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    'ttsc_rbchain_post_forward': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p,
                                            C.c_void_p, C.POINTER(Conv1dEpilogue), C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttsc_respair_supported': (C.c_int, [C.c_void_p, C.c_void_p]),
    'ttsc_respair_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p]),
    'ttsc_conv1d_destroy': (None, [C.c_void_p]),
    'ttsc_conv1d_set_weight_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttsc_conv_wgrad_grouped': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                          C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ttsc_conv_wgrad': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ttsc_conv1d_set_weight_device_dgrad': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttsc_weight_norm_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    'ttsc_weight_norm_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_int64, C.c_void_p]),
    'ttsc_matvec_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int64]),
    'ttsc_matvec': (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ttsc_rows_segment_sum': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    'ttsc_l2_normalize': (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttsc_dot_workspace_bytes': (C.c_size_t, [C.c_int64]),
    'ttsc_dot': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ttsc_div_scalar': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'ttsc_spectral_norm_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                              C.c_void_p]),
    'ttsc_adamw_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_int64, C.c_void_p]),
    'ttsc_adamw_step_guarded': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_int64, C.c_void_p, C.c_void_p]),
    'ttsc_rows_gather': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    'ttsc_rows_scatter_add': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    'ttsc_gan_loss_workspace_bytes': (C.c_size_t, [C.c_int32]),
    'ttsc_gan_loss': (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ttsc_gan_loss_lrelu': (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ttsc_bias_grad_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int64]),
    'ttsc_bias_grad': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    'ttsc_conv_wgrad_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32]),
    'ttsc_conv_wgrad_split_supported': (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'ttsc_conv_wgrad_split_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32]),
    'ttsc_conv_wgrad_split': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    'ttsc_conv_wgrad_split_grouped_supported': (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'ttsc_conv_wgrad_split_grouped_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32]),
    'ttsc_conv_wgrad_split_grouped': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                                C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t,
                                                C.c_void_p]),
    'ttsc_conv_wgrad_split_bias': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t,
                                             C.c_void_p]),
    'ttsc_deinterleave_x': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_void_p]),
    'ttsc_deinterleave_w': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    'ttsc_conv_train_supported': (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'ttsc_conv_train_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'ttsc_conv_train': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ttsc_conv_train_packed': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                         C.c_int32, C.c_void_p, C.c_void_p]),
    'ttsc_wbank_create': (C.c_int, [C.POINTER(WBankEntry), C.c_int32, C.POINTER(C.c_void_p)]),
    'ttsc_wbank_prepare': (C.c_int, [C.c_void_p, C.c_void_p]),
    'ttsc_wbank_destroy': (None, [C.c_void_p]),
    'ttsc_hifigan_create': (C.c_int, [C.POINTER(HifiganCfg), C.POINTER(C.c_void_p)]),
    'ttsc_hifigan_set_weight': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    'ttsc_hifigan_set_precision': (C.c_int, [C.c_void_p, C.c_int32]),
    'ttsc_hifigan_out_len': (C.c_int64, [C.c_void_p, C.c_int64]),
    'ttsc_hifigan_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int64]),
    'ttsc_hifigan_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    'ttsc_hifigan_forward_ragged': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_size_t, C.c_void_p]),
    'ttsc_hifigan_algorithmic_flops': (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_double)]),
    'ttsc_hifigan_destroy': (None, [C.c_void_p]),
    'ttsc_wavernn_create': (C.c_int, [C.POINTER(WavernnCfg), C.POINTER(C.c_void_p)]),
    'ttsc_wavernn_set_weight': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    'ttsc_wavernn_out_len': (C.c_int64, [C.c_void_p, C.c_int64, C.c_int64]),
    'ttsc_wavernn_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64]),
    'ttsc_wavernn_decode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                                      C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    'ttsc_wavernn_last_status': (C.c_int, [C.c_void_p, C.c_void_p]),
    'ttsc_wavernn_destroy': (None, [C.c_void_p]),
    'ttsc_linear_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                      C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    'ttsc_linear_split_supported': (C.c_int32, [C.c_int64, C.c_int32, C.c_int32, C.c_int64]),
    'ttsc_linear_forward_split': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                            C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    'ttsc_gemm_split_status': (C.c_int32, []),
    'ttsc_probe_mfma_tflops': (C.c_int, [C.c_int32, C.c_double, C.POINTER(C.c_double), C.c_void_p]),
    'ttsc_gemm_workspace_bytes': (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    'ttsc_gemm': (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                            C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ttsc_colsum_workspace_bytes': (C.c_size_t, [C.c_int64, C.c_int64]),
    'ttsc_colsum': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    'ttsc_lstm_split_status': (C.c_int32, []),
    'ttsc_split_status_stream': (C.c_int32, [C.c_void_p]),
    'ttsc_split_status_collect': (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32]),
    'ttsc_lstm_set_group_size': (C.c_int32, [C.c_int32]),
    'ttsc_melar_split_status': (C.c_int32, []),
    'ttsc_lstm_pack_whh_device': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'ttsc_lstm_seq_forward_train': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttsc_lstm_seq_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    'ttsc_gru_split_status': (C.c_int32, []),
    'ttsc_gru_pack_whh_device': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'ttsc_gru_seq_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p]),
    'ttsc_gru_seq_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_void_p]),
    'ttsc_lstm_pack_whh': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    'ttsc_lstm_seq_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    'ttsc_align_durations': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p]),
    'ttsc_expand_rows': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p]),
    'ttsc_cond_input': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    'ttsc_stft_mag': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    'ttsc_stft_mag_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    'ttsc_log_clamp': (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'ttsc_log_clamp_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'ttsc_overlap_add': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    'ttsc_device_free': (None, [C.c_void_p]),
    'ttsc_melar_create': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    'ttsc_melar_set_weights': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 11),
    'ttsc_melar_decode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    'ttsc_melar_destroy': (None, [C.c_void_p]),
}


def lib():
    """Load libttscube_hip.so once; raise (never fall back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TTSCError('HIP extension not built: %s is missing. Run `python -c "import __graft_entry__ as g; '
                        'g.build()"` or `make -C ttscube_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc, what=''):
    if rc != 0:
        msg = lib().ttsc_last_error()
        raise TTSCError('%s failed (code %d): %s' % (what, rc, msg.decode() if msg else ''))


def require_gpu():
    n = lib().ttsc_device_count()
    if n <= 0:
        raise TTSCError('no HIP device visible (ttsc_device_count=%d); the HIP path has no CPU fallback' % n)
    return n


_T = {}


def _torch_c():
    """torch._C entry points behind the two per-launch queries (resolved once): the raw current stream and the current device index straight
    from the C extension — torch.cuda.current_stream() builds a Stream object through three Python layers (~10 us; ~2300 calls and 7 ms of host
    time per Cubegan step, tools/probes/train_host_profile.py)"""
    if not _T:
        import torch
        fast = os.environ.get('TTSC_FAST_STREAM_QUERY', '1') != '0'      # (0: the torch.cuda.* path, for A/B measurements)
        _T['raw'] = getattr(torch._C, '_cuda_getCurrentRawStream', None) if fast else None
        _T['dev'] = getattr(torch._C, '_cuda_getDevice', None) if fast else None
        _T['torch'] = torch
    return _T


def current_stream():
    t = _T or _torch_c()
    if t['raw'] is not None and t['dev'] is not None:
        return C.c_void_p(t['raw'](t['dev']()))
    return C.c_void_p(t['torch'].cuda.current_stream().cuda_stream)


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(dev):
    """`with on_device(t.device):` — torch.cuda.device(dev) when `dev` is not the current device, nothing at all when it is (the usual case: one
    process per GPU)"""
    t = _T or _torch_c()
    if t['dev'] is not None and (dev.index is None or dev.index == t['dev']()):
        return _NO_GUARD
    return t['torch'].cuda.device(dev)


def dev_ptr(t):
    """Device pointer of a contiguous fp32 CUDA(HIP) tensor."""
    import torch
    assert t.is_cuda and t.is_contiguous(), 'expected a contiguous device tensor'
    return C.c_void_p(t.data_ptr())


def check_split_status(where, stream=None):
    """Raise if a multi-workgroup recurrence (split LSTM / GRU kernels) gave up on an inter-workgroup hand-off since the last
    check — its outputs are then invalid.  Synchronises the device: call once per training step / synthesis, not per layer.
    stream (a raw HIP stream handle): only the recurrences launched on that stream, waiting for that stream alone."""
    L = lib()
    if stream is not None:
        m = int(L.ttsc_split_status_stream(C.c_void_p(stream)))
        if m < 0:
            raise TTSCError('%s: ttsc_split_status_stream failed: %s' % (where, L.ttsc_last_error().decode()))
        if m:
            kinds = '/'.join(n for b, n in ((1, 'LSTM'), (2, 'GRU'), (4, 'mel-AR'), (8, 'other')) if m & b)
            raise TTSCError('%s: split %s recurrence aborted on a hand-off timeout (are other kernels occupying the CUs? '
                            'TTSC_LSTM_SPLIT=1 / TTSC_GRU_SPLIT=1 select the single-workgroup kernels)' % (where, kinds))
        return
    bad = [n for n, f in (('LSTM', L.ttsc_lstm_split_status), ('GRU', L.ttsc_gru_split_status), ('mel-AR', L.ttsc_melar_split_status)) if f() != 0]
    if bad:
        raise TTSCError('%s: split %s recurrence aborted on a hand-off timeout (are other kernels occupying the CUs? '
                        'TTSC_LSTM_SPLIT=1 / TTSC_GRU_SPLIT=1 select the single-workgroup kernels)' % (where, '/'.join(bad)))
    if L.ttsc_gemm_split_status() != 0:
        raise TTSCError('%s: an operand of a split-precision GEMM (ttsc_linear_forward_split) lay beyond the fp16 range (|v| > 65504) or was not '
                        'finite — its results are invalid; TTSC_GEMM_SPLIT=0 keeps these projections on the exact fp32 kernel' % where)


class _StagingRing:
    """page-locked staging buffers for small host-to-device uploads, reused round-robin (allocating page-locked memory per call costs more than the
    copy); a slot is reused only after the copy that last read it has completed"""
    SLOTS = 8

    def __init__(self):
        self.buf = [None] * self.SLOTS
        self.ev = [None] * self.SLOTS
        self.n = 0

    def upload(self, tensors, dev):
        """tensors: CPU int64 tensors -> device int64 views of ONE non-blocking upload, in order"""
        import torch
        total = sum(t.numel() for t in tensors)
        i = self.n % self.SLOTS
        self.n += 1
        if self.ev[i] is not None:
            self.ev[i].synchronize()
        if self.buf[i] is None or self.buf[i].numel() < total:
            self.buf[i] = torch.empty(max(total, 4096), dtype=torch.int64).pin_memory()
        host, off = self.buf[i], 0
        for t in tensors:
            host[off:off + t.numel()].copy_(t.reshape(-1))
            off += t.numel()
        d = host[:total].to(dev, non_blocking=True)
        self.ev[i] = torch.cuda.Event()
        self.ev[i].record()
        out, off = [], 0
        for t in tensors:
            out.append(d[off:off + t.numel()].view(t.shape))
            off += t.numel()
        return out


_STAGING = {}


def upload_ints(tensors, dev):
    """several small CPU int64 tensors to `dev` in one page-locked, non-blocking copy (the synthesis calls' phone ids, speaker ids and lengths: three
    uploads, two of them blocking pageable copies, were ~0.1 ms of GPU idle time in front of every sentence)"""
    import torch
    dev = torch.device(dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ring = _STAGING.get(key)
    if ring is None:
        ring = _STAGING[key] = _StagingRing()
    return ring.upload(tensors, dev)


_STATUS_WORDS = {}


def check_split_status_once(where):
    """The same verdict as check_split_status(where) for a caller whose CURRENT stream is ordered behind every stream it ran recurrences on:
    one collecting launch (ttsc_split_status_collect: every stream's recurrences + the split GEMM's range word), one 4-byte copy to page-locked
    memory and ONE wait for the current stream — instead of one blocking read-back per kernel family (four at the end of every synthesis call:
    ~100 us of a 5 ms sentence)."""
    import torch
    L = lib()
    dev = torch.cuda.current_device()
    st = _STATUS_WORDS.get(dev)
    if st is None:
        st = _STATUS_WORDS[dev] = (torch.zeros(1, dtype=torch.int32, device='cuda:%d' % dev), torch.zeros(1, dtype=torch.int32).pin_memory())
    word, host = st
    n = int(L.ttsc_split_status_collect(current_stream(), word.data_ptr(), 3))
    if n < 0:
        raise TTSCError('%s: ttsc_split_status_collect failed: %s' % (where, L.ttsc_last_error().decode()))
    if n == 0:
        return
    host.copy_(word, non_blocking=True)
    word.zero_()
    torch.cuda.current_stream().synchronize()
    m = int(host[0])
    if m & 15:
        kinds = '/'.join(k for b, k in ((1, 'LSTM'), (2, 'GRU'), (4, 'mel-AR'), (8, 'other')) if m & b)
        raise TTSCError('%s: split %s recurrence aborted on a hand-off timeout (are other kernels occupying the CUs? '
                        'TTSC_LSTM_SPLIT=1 / TTSC_GRU_SPLIT=1 select the single-workgroup kernels)' % (where, kinds))
    if m & 16:
        raise TTSCError('%s: an operand of a split-precision GEMM (ttsc_linear_forward_split) lay beyond the fp16 range (|v| > 65504) or was not '
                        'finite — its results are invalid; TTSC_GEMM_SPLIT=0 keeps these projections on the exact fp32 kernel' % where)


class lstm_group_size:
    """`with lstm_group_size(8): ...` — utterances per member group of the split LSTM recurrences inside the block (ttsc_lstm_set_group_size:
    fewer CUs held per padded batch, same results); the previous setting is restored on exit."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.prev = int(lib().ttsc_lstm_set_group_size(self.n))
        if self.prev < 0:
            raise TTSCError('ttsc_lstm_set_group_size(%r): not one of 0, 1, 2, 4, 8' % (self.n,))
        return self

    def __exit__(self, *exc):
        lib().ttsc_lstm_set_group_size(self.prev)
        return False


class DevLengths(list):
    """A list of per-utterance lengths that also carries its int32 device copy (`.dev`): the host list keeps every existing use working, the
    kernels take `.dev` — ONE host-to-device copy per synthesis call instead of one synchronous pageable copy (each a stream drain) per layer."""

    def __init__(self, values, dev_tensor=None, device=None):
        super().__init__(int(v) for v in values)
        if dev_tensor is None:
            import torch
            dev_tensor = torch.tensor(list(self), dtype=torch.int32).pin_memory().to(device, non_blocking=True)
        self.dev = dev_tensor


def lengths_dev(lengths, device):
    """int32 device tensor of `lengths` (a DevLengths' own copy, a device tensor as is, else a fresh — synchronous — upload)"""
    import torch
    if lengths is None:
        return None
    if isinstance(lengths, DevLengths) and lengths.dev.device == torch.device(device):
        return lengths.dev
    if torch.is_tensor(lengths) and lengths.is_cuda:
        return lengths.to(torch.int32).contiguous()
    return torch.as_tensor(lengths, dtype=torch.int32, device=device).contiguous()
