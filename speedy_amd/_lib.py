"""Loader for libspeedy_hip.so.  Fails loudly: the product has no CPU path."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPEEDY_HIP_LIB") or os.path.join(_HERE, "lib", "libspeedy_hip.so")
_LIB = None

c_short_p = C.POINTER(C.c_short)
c_float_p = C.POINTER(C.c_float)
c_int64_p = C.POINTER(C.c_int64)


class StreamJob(C.Structure):  # spx_stream_job, include/speedy_hip.h
    _fields_ = [("in_off", C.c_int64), ("n_in", C.c_int64), ("out_off", C.c_int64), ("out_cap", C.c_int64),
                ("channels", C.c_int32), ("speed", C.c_float), ("nonlinear", C.c_float), ("feedback", C.c_float)]


class Taps(C.Structure):  # spx_taps
    _fields_ = [("tension", C.c_void_p), ("speed", C.c_void_p), ("features", C.c_void_p),
                ("spectrogram", C.c_void_p), ("normalized", C.c_void_p)]


TENSION_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_float)
FEATURES_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, c_float_p)

# name -> (restype, argtypes); every symbol declared in include/speedy_hip.h and include/sonic2.h
SYMBOLS = {
    "spx_last_error": (C.c_char_p, []),
    "spx_abi_version": (C.c_int, []),
    "spx_plan_create": (C.c_void_p, [C.c_int, C.c_int]),
    "spx_plan_destroy": (None, [C.c_void_p]),
    "spx_plan_frame_step": (C.c_int, [C.c_void_p]),
    "spx_plan_window_size": (C.c_int, [C.c_void_p]),
    "spx_plan_fft_size": (C.c_int, [C.c_void_p]),
    "spx_plan_future": (C.c_int, [C.c_void_p]),
    "spx_plan_max_required": (C.c_int, [C.c_void_p]),
    "spx_plan_frames": (C.c_int64, [C.c_void_p, C.c_int64]),
    "spx_plan_out_capacity": (C.c_int64, [C.c_void_p, C.c_int64, C.c_float]),
    "spx_plan_out_capacity_for": (C.c_int64, [C.c_void_p, C.c_int64, C.c_float, C.c_float]),
    "spx_batch_workspace_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(StreamJob), C.c_int]),
    "spx_batch_run": (C.c_int, [C.c_void_p, C.POINTER(StreamJob), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_size_t, C.POINTER(Taps), C.c_void_p]),
    "spx_batch_run_ahead": (C.c_int, [C.c_void_p, C.POINTER(StreamJob), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.POINTER(Taps), C.c_void_p]),
    "spx_batch_run_overlapped": (C.c_int, [C.c_void_p, C.POINTER(StreamJob), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.POINTER(Taps), C.c_void_p]),
    "spx_batch_run_ahead_when": (C.c_int, [C.c_void_p, C.POINTER(StreamJob), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.POINTER(Taps), C.c_void_p, C.c_void_p]),
    "spx_batch_workspace_bytes_mixed": (C.c_size_t, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(StreamJob), C.POINTER(C.c_int), C.c_int]),
    "spx_batch_run_mixed": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(StreamJob), C.POINTER(C.c_int), C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "spx_batch_run_mixed_ahead": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(StreamJob), C.POINTER(C.c_int), C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "spx_batch_run_mixed_taps": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(StreamJob), C.POINTER(C.c_int), C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Taps), C.c_void_p]),
    "spx_batch_read_steps": (C.c_int, [C.c_void_p, C.POINTER(StreamJob), C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "spx_batch_read_steps_mixed": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(StreamJob), C.POINTER(C.c_int), C.c_int,
                                             C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "spx_batch_analyze": (C.c_int, [C.c_void_p, C.POINTER(StreamJob), C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.POINTER(Taps), C.c_void_p]),
    "spx_batch_walk": (C.c_int, [C.c_void_p, C.POINTER(StreamJob), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_size_t, C.POINTER(Taps), C.c_void_p]),
    "spx_set_timing": (None, [C.c_int]),
    "spx_set_pipeline_chunks": (None, [C.c_int]),
    "spx_set_concurrent": (None, [C.c_int]),
    "spx_timing_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "spx_timing_last_tension_ms": (C.c_double, []),
    "spx_batch_kernel_names": (C.c_char_p, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "spx_batch_kernel_names_lean": (C.c_char_p, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "spx_debug_last_walk_form": (C.c_int, []),
    "spx_debug_kernel_vgprs": (C.c_int, [C.c_int]),
    "spx_debug_last_call_concurrent": (C.c_int, []),
    "spx_debug_fdiv_check": (C.c_longlong, [C.c_uint, C.c_uint, C.c_int, C.c_int]),
    "spx_debug_xfade_check": (C.c_longlong, [C.c_int, C.c_int]),
    "spx_debug_arith_check": (C.c_longlong, [C.c_uint, C.c_uint, C.c_uint]),
    "spx_debug_log_check": (C.c_int, [C.c_uint, C.c_uint, C.POINTER(C.c_ulonglong)]),
    "spx_debug_walk_info": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "spx_debug_analysis_info": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "spx_debug_mode_resources": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]),
    "spx_pipeline_create": (C.c_void_p, [C.c_void_p, C.POINTER(StreamJob), C.c_int, C.c_int, C.c_uint]),
    "spx_pipeline_create_mixed": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(StreamJob), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_uint]),
    "spx_pipeline_destroy": (None, [C.c_void_p]),
    "spx_pipeline_depth": (C.c_int, [C.c_void_p]),
    "spx_pipeline_input_values": (C.c_size_t, [C.c_void_p]),
    "spx_pipeline_host_input": (C.c_void_p, [C.c_void_p]),
    "spx_pipeline_submit": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int]),
    "spx_pipeline_input_consumed": (C.c_int, [C.c_void_p, C.c_int64]),
    "spx_pipeline_wait": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "spx_host_alloc": (C.c_void_p, [C.c_size_t]),
    "spx_host_free": (None, [C.c_void_p]),
    "spx_device_alloc": (C.c_void_p, [C.c_size_t]),
    "spx_device_free": (None, [C.c_void_p]),
    "spx_copy_to_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "spx_copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "spx_batch_pack_outputs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spx_stream_synchronize": (C.c_int, [C.c_void_p]),
    # include/sonic2.h
    "sonicCreateStream": (C.c_void_p, [C.c_int, C.c_int]),
    "sonicDestroyStream": (None, [C.c_void_p]),
    "sonicWriteShortToStream": (C.c_int, [C.c_void_p, c_short_p, C.c_int]),
    "sonicReadShortFromStream": (C.c_int, [C.c_void_p, c_short_p, C.c_int]),
    "sonicWriteFloatToStream": (C.c_int, [C.c_void_p, c_float_p, C.c_int]),
    "sonicReadFloatFromStream": (C.c_int, [C.c_void_p, c_float_p, C.c_int]),
    "sonicSetRate": (None, [C.c_void_p, C.c_float]),
    "sonicSetSpeed": (None, [C.c_void_p, C.c_float]),
    "sonicFlushStream": (C.c_int, [C.c_void_p]),
    "sonicEnableNonlinearSpeedup": (None, [C.c_void_p, C.c_float]),
    "sonicSetDurationFeedbackStrength": (None, [C.c_void_p, C.c_float]),
    "getSonicBufferSize": (C.c_int, [C.c_void_p]),
    "sonicSpectrogramSize": (C.c_int, [C.c_void_p]),
    "sonicTensionCallback": (None, [C.c_void_p, TENSION_FN]),
    "getSonicTensionCallback": (C.c_void_p, [C.c_void_p]),
    "sonicSpeedCallback": (None, [C.c_void_p, TENSION_FN]),
    "getSonicSpeedCallback": (C.c_void_p, [C.c_void_p]),
    "sonicFeaturesCallback": (None, [C.c_void_p, FEATURES_FN]),
    "getSonicFeaturesCallback": (C.c_void_p, [C.c_void_p]),
    "sonicSpectrogramCallback": (None, [C.c_void_p, FEATURES_FN]),
    "getSonicSpectrogramCallback": (C.c_void_p, [C.c_void_p]),
    "sonicNormalizedSpectrogramCallback": (None, [C.c_void_p, FEATURES_FN]),
    "getSonicNormalizedSpectrogramCallback": (C.c_void_p, [C.c_void_p]),
    "sonicIntGetNumChannels": (C.c_int, [C.c_void_p]),
    "sonicIntGetSampleRate": (C.c_int, [C.c_void_p]),
    "sonicIntGetSpeed": (C.c_float, [C.c_void_p]),
    "sonicIntCreateStream": (C.c_void_p, [C.c_int, C.c_int]),
    "sonicIntDestroyStream": (None, [C.c_void_p]),
    "sonicIntSetSpeed": (None, [C.c_void_p, C.c_float]),
    "sonicIntSetRate": (None, [C.c_void_p, C.c_float]),
    "sonicIntWriteShortToStream": (C.c_int, [C.c_void_p, c_short_p, C.c_int]),
    "sonicIntWriteFloatToStream": (C.c_int, [C.c_void_p, c_float_p, C.c_int]),
    "sonicIntReadShortFromStream": (C.c_int, [C.c_void_p, c_short_p, C.c_int]),
    "sonicIntReadFloatFromStream": (C.c_int, [C.c_void_p, c_float_p, C.c_int]),
    "sonicIntFlushStream": (C.c_int, [C.c_void_p]),
    "sonicIntSamplesAvailable": (C.c_int, [C.c_void_p]),
    "sonicIntSetUserData": (None, [C.c_void_p, C.c_void_p]),
    "sonicIntGetUserData": (C.c_void_p, [C.c_void_p]),
    # include/speedy.h
    "speedyCreateStream": (C.c_void_p, [C.c_int]),
    "speedyDestroyStream": (None, [C.c_void_p]),
    "speedyInputFrameSize": (C.c_int, [C.c_void_p]),
    "speedyInputFrameStep": (C.c_int, [C.c_void_p]),
    "speedyFFTSize": (C.c_int, [C.c_void_p]),
    "speedyBinToFreq": (C.c_float, [C.c_void_p, C.c_int]),
    "speedyFreqToBin": (C.c_int, [C.c_void_p, C.c_float]),
    "speedyAddData": (None, [C.c_void_p, c_float_p, C.c_int64]),
    "speedyAddDataShort": (None, [C.c_void_p, c_short_p, C.c_int64]),
    "speedyComputeTension": (C.c_int, [C.c_void_p, C.c_int64, c_float_p]),
    "speedyComputeSpeedFromTension": (C.c_float, [C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "speedyGetCurrentTime": (C.c_int64, [C.c_void_p]),
    "speedySpectrogram": (c_float_p, [C.c_void_p, c_float_p]),
    "speedyGetSpectrogram": (c_float_p, [C.c_void_p]),
    "speedyGetSpectrogramAtTime": (c_float_p, [C.c_void_p, C.c_int64]),
    "speedyGetNormalizedSpectrogram": (c_float_p, [C.c_void_p]),
    "speedyGetInternalState": (c_float_p, [C.c_void_p]),
    "speedyGetEnergyCompressed": (C.c_float, [C.c_void_p]),
    "speedyGetSpeechChanges": (C.c_float, [C.c_void_p]),
    "speedyHipHysteresisFuture": (C.c_int, [C.c_void_p]),
    "speedyHipHysteresisPast": (C.c_int, [C.c_void_p]),
    # the hooks between stages (speedy.h:102-133)
    "speedyComputeSpectralDifference": (None, [C.c_void_p, c_float_p, c_float_p, C.c_int64]),
    "speedyComputeLocalEnergy": (None, [C.c_void_p, c_float_p, C.c_int64]),
    "speedySaveSpectrogramData": (None, [C.c_void_p, c_float_p, C.c_int64]),
    "speedyPreemphasisFilter": (None, [C.c_void_p, c_float_p, C.c_int]),
    "speedyEvaluateHysteresis": (C.c_float, [C.c_void_p, C.c_int64]),
    "speedyAddToHysteresisBuffer": (None, [C.c_void_p, C.c_float, C.c_int64]),
    "speedyGetInternalSpectrogram": (c_float_p, [C.c_void_p]),
    "speedyGetInternalNormalizedSpectrogram": (c_float_p, [C.c_void_p]),
    "speedyNormalizeByEnergy": (C.c_float, [c_float_p, c_float_p, C.c_int]),
    "CreateFirstOrderFilter": (C.c_void_p, [C.c_float]),
    "DesignFirstOrderLowpassFilter": (None, [C.c_void_p, C.c_float]),
    "IterateFirstOrderFilter": (C.c_float, [C.c_void_p, C.c_float]),
    "ResetFirstOrderFilter": (None, [C.c_void_p]),
    "DeleteFirstOrderFilter": (None, [C.c_void_p]),
    "speedyHipSetMatchMatlab": (None, [C.c_int]),
    "speedyHipCreateSonicStream": (C.c_void_p, [C.c_int, C.c_int, C.c_int]),
    "speedyHipSetCoalescing": (None, [C.c_int]),
    "speedyHipGetCoalescing": (C.c_int, []),
    "speedyHipCreateSonicStreamEx": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "speedyHipPoolStats": (None, [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "sonicSamplesAvailable": (C.c_int, [C.c_void_p]),
    "speedyHipLastError": (C.c_char_p, []),
}


def build():
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def lib():
    """The loaded library.  torch is imported first so that one HIP runtime (torch's) serves both."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the product has no CPU fallback)")
    import torch  # noqa: F401  (loads libamdhip64 with the soname the library links against)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        if os.environ.get("SPEEDY_HIP_LIB") and not hasattr(L, name):
            continue           # a developer's A/B build of an older tree (tools/build_variant.sh) may lack newer diagnostics
        fn = getattr(L, name)  # AttributeError = the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L
