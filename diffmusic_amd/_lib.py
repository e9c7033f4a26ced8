"""ctypes binding of libdiffmusic_hip.so (C ABI in include/diffmusic_hip.h).

The product path has no CPU fallback: if the library is missing or fails to load this module
raises immediately."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMX_LIB_PATH") or os.path.join(_HERE, "lib", "libdiffmusic_hip.so")   # DMX_LIB_PATH: dev A/B builds
MAX_STAGES = 8


class HifiGanConfig(C.Structure):
    _fields_ = [("model_in_dim", C.c_int), ("upsample_initial_channel", C.c_int), ("num_upsamples", C.c_int),
                ("upsample_rates", C.c_int * MAX_STAGES), ("upsample_kernel_sizes", C.c_int * MAX_STAGES),
                ("num_kernels", C.c_int), ("resblock_kernel_sizes", C.c_int * MAX_STAGES),
                ("num_dilations", C.c_int), ("resblock_dilation_sizes", C.c_int * (MAX_STAGES * MAX_STAGES)),
                ("leaky_relu_slope", C.c_float)]


class VaeConfig(C.Structure):
    _fields_ = [("latent_channels", C.c_int), ("out_channels", C.c_int), ("num_blocks", C.c_int),
                ("block_out_channels", C.c_int * MAX_STAGES), ("layers_per_block", C.c_int),
                ("norm_num_groups", C.c_int), ("eps", C.c_float)]


class UNetConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("num_blocks", C.c_int),
                ("block_out_channels", C.c_int * MAX_STAGES), ("layers_per_block", C.c_int),
                ("attention_heads", C.c_int), ("norm_num_groups", C.c_int),
                ("down_attn", C.c_int * MAX_STAGES), ("up_attn", C.c_int * MAX_STAGES),
                ("class_embed_dim", C.c_int), ("num_attn_per_layer", C.c_int), ("attn_cross_dims", C.c_int * 4)]


class HtsatConfig(C.Structure):
    _fields_ = [("spec_size", C.c_int), ("num_mel_bins", C.c_int), ("patch_size", C.c_int), ("embed_dim", C.c_int), ("window_size", C.c_int),
                ("num_stages", C.c_int), ("depths", C.c_int * 4), ("num_heads", C.c_int * 4), ("ln_eps", C.c_float), ("bn_eps", C.c_float)]


class GemmDesc(C.Structure):
    """Mirror of csrc/dmx_common.h::GemmDesc (test hook dmx_gemm_raw only)."""
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("C2", C.c_void_p),
                ("bias", C.c_void_p), ("rowbias", C.c_void_p), ("R", C.c_void_p), ("X", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ldw", C.c_int),
                ("Hi", C.c_int), ("Wi", C.c_int), ("Ci", C.c_int), ("lda", C.c_int),
                ("Hq", C.c_int), ("Wq", C.c_int), ("sy", C.c_int), ("sx", C.c_int),
                ("ntaps", C.c_int),
                ("Ho", C.c_int), ("Wo", C.c_int), ("ldc", C.c_int), ("osy", C.c_int), ("ooy", C.c_int),
                ("osx", C.c_int), ("oox", C.c_int),
                ("ldr", C.c_int), ("ldx", C.c_int), ("ldc2", C.c_int),
                ("Z", C.c_int), ("Zi", C.c_int),
                ("sAo", C.c_longlong), ("sAi", C.c_longlong), ("sWo", C.c_longlong), ("sWi", C.c_longlong),
                ("sCo", C.c_longlong), ("sCi", C.c_longlong),
                ("alpha", C.c_float), ("act_slope", C.c_float), ("mask_slope", C.c_float),
                ("flags", C.c_int),
                ("tdy", C.c_byte * 16), ("tdx", C.c_byte * 16), ("resid_inv_slope", C.c_float), ("tile_cfg", C.c_int), ("ldrb", C.c_int), ("ksplit", C.c_int),
                ("XB", C.c_void_p), ("B2", C.c_void_p), ("ldxb", C.c_int), ("ldb2", C.c_int),
                ("colsum", C.c_void_p), ("ln_eps", C.c_float), ("rowstats_in", C.c_void_p), ("rowstats_out", C.c_void_p), ("nslots", C.c_int),
                ("gn_part", C.c_void_p), ("gnb_x", C.c_void_p), ("gnb_scale", C.c_void_p), ("gnb_shift", C.c_void_p),
                ("gnb_ldx", C.c_int), ("gnb_silu", C.c_int), ("gnb_stats", C.c_void_p), ("gnb_cpg", C.c_int)]


EPI_BIAS, EPI_ROWBIAS, EPI_RESID, EPI_ACCUM, EPI_MASK, EPI_LRELU2, EPI_TANH, EPI_F32OUT, EPI_NO_C, EPI_RESID_INV, EPI_MASKBITS, EPI_BITS2, EPI_SOFTBWD = \
    1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096
EPI_GEGLU, EPI_LNFOLD, EPI_ROWSTATS, EPI_GNSTATS, EPI_GNBWD = 8192, 16384, 32768, 65536, 131072

_SIGS = {
    "dmx_abi_version": (C.c_int, []),
    "dmx_act_dtype": (C.c_int, []),
    "dmx_last_error": (C.c_char_p, []),
    "dmx_hifigan_create": (C.c_void_p, [C.POINTER(HifiGanConfig)]),
    "dmx_vae_decoder_create": (C.c_void_p, [C.POINTER(VaeConfig)]),
    "dmx_unet_create": (C.c_void_p, [C.POINTER(UNetConfig)]),
    "dmx_model_destroy": (None, [C.c_void_p]),
    "dmx_model_num_params": (C.c_int, [C.c_void_p]),
    "dmx_model_param_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "dmx_model_param_numel": (C.c_size_t, [C.c_void_p, C.c_int]),
    "dmx_model_param_ndim": (C.c_int, [C.c_void_p, C.c_int]),
    "dmx_model_param_dim": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "dmx_model_load_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "dmx_model_finalize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dmx_hifigan_out_len": (C.c_int, [C.c_void_p, C.c_int]),
    "dmx_hifigan_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "dmx_hifigan_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmx_hifigan_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dmx_vae_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dmx_vae_decode_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmx_vae_decode_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "dmx_unet_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dmx_unet_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmx_unet_fwd_ctx": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmx_unet_workspace_bytes_ctx": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dmx_htsat_create": (C.c_void_p, [C.POINTER(HtsatConfig)]),
    "dmx_htsat_feature_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dmx_htsat_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "dmx_htsat_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmx_htsat_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dmx_htsat_tape_raw": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmx_gram_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dmx_gram_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dmx_gemm_raw": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmx_prof_begin": (None, []),
    "dmx_prof_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dmx_flash_attn_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_float, C.c_void_p]),
    "dmx_gemm_splitk_workspace": (C.c_int, [C.c_void_p, C.c_size_t]),
    "dmx_conv_pair_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmx_conv_pair_group_raw": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dmx_groupnorm_scratch_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "dmx_groupnorm_raw": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_void_p]),
    "dmx_groupnorm_part_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "dmx_groupnorm_parts_raw": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dmx_gemm_last_tile_rows_raw": (C.c_int, []),
    "dmx_groupnorm_bwd_raw": (C.c_int, [C.c_void_p] * 10 + [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dmx_prof_dominant": (C.c_int, [C.POINTER(C.c_double)] * 3),
    "dmx_audio_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dmx_audio_destroy": (None, [C.c_void_p]),
    "dmx_audio_num_frames": (C.c_int, [C.c_void_p, C.c_int]),
    "dmx_audio_num_bins": (C.c_int, [C.c_void_p]),
    "dmx_audio_state_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "dmx_audio_transform_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "dmx_audio_transform_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "dmx_audio_is_fused": (C.c_int, [C.c_void_p, C.c_int]),
    "dmx_audio_guidance_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "dmx_audio_guidance_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_void_p,
                                         C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                         C.c_void_p]),
    "dmx_audio_stft_mag": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dmx_audio_stft_mag_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dmx_audio_melscale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "dmx_mask_apply": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dmx_l2_loss": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_void_p]),
    "dmx_grad_normalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_void_p]),
    "dmx_fir_fwd": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong] + [C.c_int] * 7 + [C.c_void_p]),
    "dmx_fir_bwd": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong] + [C.c_int] * 7 + [C.c_void_p]),
    "dmx_sched_pred_x0": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_void_p]),
    "dmx_sched_pred_x0_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_int, C.c_float, C.c_void_p]),
    "dmx_sched_step_ex": (C.c_int, [C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_int] + [C.c_float] * 5 + [C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "dmx_sched_cfg_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_void_p]),
    "dmx_randn_philox": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.POINTER(C.c_ulonglong), C.c_ulonglong, C.c_void_p]),
    "dmx_sched_step": (C.c_int, [C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_int] + [C.c_float] * 5 + [C.c_int, C.c_void_p]),
}

_lib = None


def lib():
    """Returns the loaded library; raises (no fallback) if it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m diffmusic_amd.build` "
                               "(the diffmusic_amd hot path has no CPU fallback)")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if not hasattr(h, name):
                continue               # optional symbols are checked by tests/test_abi.py
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        if h.dmx_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libdiffmusic_hip.so ABI version {h.dmx_abi_version()} != {ABI_VERSION} expected by this package: rebuild it "
                               "(python -m diffmusic_amd.build --force)")
        _lib = h
    return _lib


ABI_VERSION = 4          # include/diffmusic_hip.h DMX_ABI_VERSION


def act_dtype():
    """torch dtype of the library's 16-bit activation tensors (fp16 unless built with -DDMX_BF16)."""
    import torch
    return torch.float16 if lib().dmx_act_dtype() == 1 else torch.bfloat16


class DmxError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = lib().dmx_last_error()
        raise DmxError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
