"""ctypes binding of libbmnas_hip.so (C ABI: include/bmnas_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is
absent, importing/using this module raises.  Every wrapper takes torch CUDA(HIP) fp32
contiguous tensors, passes raw device pointers + the current torch stream, and turns a
non-zero return code into an exception.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('BMNAS_LIB') or os.path.join(HERE, 'libbmnas_hip.so')   # BMNAS_LIB: A/B runs of two builds
MAX_PTRS = 16


class BmnasError(RuntimeError):
    pass


_NOTED = set()


def note_off_path(where, why):
    """A caller-side module (classifier, criterion, reshape layer) was given something the gfx950
    kernels do not cover and is about to run its torch parent instead.  The fusion cell itself
    never does this (it raises); here the parent class IS the reference behaviour, so the call
    proceeds — but never silently: one warning per (where, why)."""
    key = (where, why)
    if key not in _NOTED:
        _NOTED.add(key)
        import warnings
        warnings.warn(f'bmnas: {where} runs on the stock torch ops, not the HIP kernels: {why}', RuntimeWarning,
                      stacklevel=3)


class Dropout(C.Structure):
    """bmnas_dropout_t"""
    _fields_ = [('thr', C.c_uint32), ('scale', C.c_float), ('seed', C.c_uint64),
                ('offset', C.c_uint64), ('step', C.c_void_p)]


NO_DROP = Dropout(0, 1.0, 0, 0, None)


class MixEp(C.Structure):
    """bmnas_mix_ep_t"""
    _fields_ = [('U', C.c_void_p), ('chan', C.c_void_p), ('x', C.c_void_p), ('p1', C.c_void_p),
                ('gamma', C.c_void_p), ('dgamma', C.c_void_p), ('dgamma_shards', C.c_int),
                ('dgamma_shard_stride', C.c_int64), ('dx', C.c_void_p), ('accumulate_dx', C.c_int),
                ('dV', C.c_void_p), ('bn_grad', C.c_void_p), ('q', C.c_int), ('drop_glu', Dropout),
                ('drop_fc', Dropout)]


class BnFin(C.Structure):
    """bmnas_bn_fin_t"""
    _fields_ = [('stat', C.c_void_p), ('conv_bias', C.c_void_p), ('bn_w', C.c_void_p), ('bn_b', C.c_void_p),
                ('running_mean', C.c_void_p), ('running_var', C.c_void_p), ('num_batches_tracked', C.c_void_p),
                ('shards', C.c_int), ('n_nbt', C.c_int), ('training', C.c_int), ('on', C.c_int)]


NO_FIN = BnFin()


class LazyLn(C.Structure):
    """bmnas_lazy_ln_t: a step-node output whose LayerNorm its consumers apply (csrc/lazyln.hip)"""
    _fields_ = [('pre', C.c_void_p), ('rec', C.c_void_p), ('prm', C.c_void_p), ('ln_w', C.c_void_p),
                ('ln_b', C.c_void_p), ('stats', C.c_void_p)]

MAX_GROUP = 8


class PoolProb(C.Structure):
    """bmnas_pool_prob_t"""
    _fields_ = [('x', C.c_void_p), ('out', C.c_void_p), ('idx', C.c_void_p), ('g', C.c_void_p), ('dx', C.c_void_p),
                ('C', C.c_int), ('H', C.c_int), ('W', C.c_int), ('oh', C.c_int), ('ow', C.c_int)]


class ConvFwdProb(C.Structure):
    """bmnas_conv_fwd_prob_t"""
    _fields_ = [('src', C.c_void_p), ('W', C.c_void_p), ('bias', C.c_void_p), ('U', C.c_void_p),
                ('stat', C.c_void_p), ('C_in', C.c_int), ('ldw', C.c_int)]


class BnReluFwdProb(C.Structure):
    """bmnas_bn_relu_fwd_prob_t"""
    _fields_ = [('U', C.c_void_p), ('chan', C.c_void_p), ('out', C.c_void_p), ('fin', BnFin), ('drop', Dropout)]


class BnReluBwdProb(C.Structure):
    """bmnas_bn_relu_bwd_prob_t"""
    _fields_ = [('g', C.c_void_p), ('U', C.c_void_p), ('chan', C.c_void_p), ('dV', C.c_void_p),
                ('bn_grad', C.c_void_p), ('drop', Dropout)]


class ConvBwdProb(C.Structure):
    """bmnas_conv_bwd_prob_t"""
    _fields_ = [('dV', C.c_void_p), ('W', C.c_void_p), ('src', C.c_void_p), ('dsrc', C.c_void_p),
                ('dW', C.c_void_p), ('dbias', C.c_void_p), ('bn_U', C.c_void_p), ('bn_chan', C.c_void_p),
                ('bn_grad', C.c_void_p), ('C_in', C.c_int), ('ldw', C.c_int), ('ldw_grad', C.c_int),
                ('accumulate', C.c_int)]


def make_bn_fin(stat, shards, conv_bias, bn_w, bn_b, rm, rv, nbt, training):
    """Descriptor for in-kernel BatchNorm finalisation (bmnas_bn_fin_t): the consumer of a conv output
    derives scale / shift from the atomically accumulated batch sums `stat` (training) or from the
    running statistics (eval) and writes `chan` for the backward kernels."""
    p = lambda t: None if t is None else t.data_ptr()
    return BnFin(p(stat), p(conv_bias), p(bn_w), p(bn_b), p(rm), p(rv), p(nbt), int(shards),
                 0 if nbt is None else nbt.numel(), int(training), 1)

_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_I = C.c_int
_I64 = C.c_int64
_U32 = C.c_uint32

SIGNATURES = {
    'bmnas_version': ([], _I),
    'bmnas_dropout_mask': ([Dropout, _I64, _P, _P], _I),
    'bmnas_mixsum_fwd': ([_PP, _I, _P, _I, _P, _I64, _P], _I),
    'bmnas_mixsum_bwd': ([_PP, _PP, _I, _P, _I, _P, _P, _P, _I, _I64, _U32, _I64, _P], _I),
    'bmnas_mixsum_pair_fwd': ([_PP, _I, _P, _I, _P, _I, _P, _P, _I64, _P], _I),
    'bmnas_mixsum_pair_bwd': ([_PP, _PP, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I64, _U32, _I64, _P],
                              _I),
    'bmnas_cat_ln_fwd': ([_PP, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P], _I),
    'bmnas_cat_ln_bwd': ([_P, _PP, _I, _P, _P, _P, _P, _PP, _P, _U32, _P, _P, _I, _I, _I, _I, _P, _I64, _P], _I),
    'bmnas_ln_affine_bwd': ([_P, _P, _PP, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    'bmnas_ln_affine_bwd_multi': ([_I, _PP, _PP, C.POINTER(_PP), C.POINTER(C.c_int), _PP, _PP, _PP, _PP, _PP,
                                  _PP, _I, C.POINTER(C.c_int), _I, C.POINTER(C.c_int), C.POINTER(C.c_int), _P],
                                 _I),
    'bmnas_sdpa_ln_fwd': ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, Dropout, _P], _I),
    'bmnas_sdpa_ln_bwd': ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _U32, _I, _I, _I, Dropout, _P], _I),
    'bmnas_conv1x1_num_partials': ([_I, _I], _I),
    'bmnas_conv1x1_fwd': ([_PP, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    'bmnas_conv1x1_bwd_data': ([_P, _P, _I, _I, _PP, _I, _I, _U32, _I, _I, _I, _P], _I),
    'bmnas_conv1x1_fwd_sdpa': ([_PP, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I,
                                _P, _P, _P, _P, _P, _P, _P, _I, Dropout, _P], _I),
    'bmnas_conv1x1_bwd_all_sdpa': ([_P, _P, _I, _I, _PP, _I, _I, _U32, _I, _I, _I, _PP, _P, _I, _P, _I,
                                    _P, _P, _P, _P, _P, _P, _P, _P, _P, _U32, _I, Dropout, _P, _P, _P, _I, _P], _I),
    'bmnas_conv1x1_bwd_all': ([_P, _P, _I, _I, _PP, _I, _I, _U32, _I, _I, _I, _PP, _P, _I, _P, _I,
                               _P, _P, _P, _I, _P], _I),
    'bmnas_conv1x1_bwd_all_mix_ok': ([_I, _I, _I, _I, _I], _I),
    'bmnas_conv1x1_bwd_all_mix': ([_P, _P, _I, _I, _PP, _I, _I, _U32, _I, _I, _I, _PP, _P, _I, _P, _I,
                                   _P, _P, _P, _I, C.POINTER(MixEp), _P], _I),
    'bmnas_conv1x1_bwd_weight': ([_P, _PP, _I, _I, _P, _I, _P, _I, _I, _I, _I, _P], _I),
    'bmnas_fold_weight': ([_P, _P, _I, _I, _P], _I),
    'bmnas_probe_read': ([_P, _I64, _I, _I, _P, _P], _I),
    'bmnas_probe_barrier': ([_P, _P, _P, _I64, _I, _I, _P, _I, _P], _I),
    'bmnas_conv_family_calls': ([C.POINTER(C.c_long), _I, _I], _I),
    'bmnas_bn_finalize': ([_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P], _I),
    'bmnas_node_mix_fwd': ([_P, _P, _P, _P, _P, BnFin, _P, _P, _I, _I, _I, Dropout, Dropout, _P], _I),
    'bmnas_node_mix_conv_fwd_ok': ([_I, _I, _I, _I], _I),
    'bmnas_node_mix_conv_fwd': ([_P, _P, _P, _P, _P, BnFin, _P, _P, Dropout, Dropout, _PP, _I, _P, _I, _P, _P, _P, _I,
                                 _I, _I, _I, _P], _I),
    'bmnas_node_mix_fwd_next': ([_P, _P, _P, _P, _P, BnFin, _P, _P, _I, _I, _I, Dropout, Dropout, _PP, _I, _P, _I,
                                 _P, _P], _I),
    'bmnas_node_mix_bwd_next': ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _P, _P, _U32, _P, _P, _I, _I, _I,
                                 Dropout, Dropout, _PP, _PP, _I, _U32, _P, _I, _P, _I, _I64, _P, _P, _P, _P, _P], _I),
    'bmnas_node_mix_ln_fwd': ([_P, _P, _P, _P, _P, BnFin, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, Dropout,
                               Dropout, _P, _P], _I),
    'bmnas_node_mix_bwd': ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _P, _P, _U32, _P, _P, _I, _I, _I,
                            Dropout, Dropout, _P], _I),
    'bmnas_node_mix_ln_bwd_ok': ([_I, _I, _I], _I),
    'bmnas_node_mix_ln_bwd': ([_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _P, _P, _U32,
                               _P, _P, _I, _I, _I, Dropout, Dropout, _P], _I),
    'bmnas_bn_glu_fwd': ([_P, _P, BnFin, _P, _I, _I, _I, Dropout, _P], _I),
    'bmnas_bn_glu_bwd': ([_P, _P, _P, _P, _P, _I, _I, _I, Dropout, _P], _I),
    'bmnas_bn_relu_fwd': ([_P, _P, BnFin, _P, _I, _I, _I, Dropout, _P], _I),
    'bmnas_bn_relu_bwd': ([_P, _P, _P, _P, _P, _I, _I, _I, Dropout, _P], _I),
    'bmnas_adaptive_maxpool_fwd_group': ([C.POINTER(PoolProb), _I, _I, _P], _I),
    'bmnas_adaptive_maxpool_bwd_group': ([C.POINTER(PoolProb), _I, _I, _P], _I),
    'bmnas_conv1x1_group_ok': ([_I, C.POINTER(C.c_int), _I, _I, _I], _I),
    'bmnas_conv1x1_fwd_group': ([C.POINTER(ConvFwdProb), _I, _I, _I, _I, _I, _P], _I),
    'bmnas_bn_relu_fwd_group': ([C.POINTER(BnReluFwdProb), _I, _I, _I, _I, _P], _I),
    'bmnas_bn_relu_bwd_group': ([C.POINTER(BnReluBwdProb), _I, _I, _I, _I, _P], _I),
    'bmnas_conv1x1_bwd_group': ([C.POINTER(ConvBwdProb), _I, _I, _I, _I, _I, _P], _I),
    'bmnas_bn_relu_ln_fwd': ([_P, _P, BnFin, _P, _P, _P, _P, _P, _P, _I, _I, _I, Dropout, _P, _P], _I),
    'bmnas_bn_relu_ln_fwd_pair_ok': ([_I, _I, _I, _I], _I),
    'bmnas_bn_relu_ln_fwd_pair': ([_P, _P, BnFin, _P, _P, _P, _P, _P, _P, _I, _I, _I, Dropout, _P, _PP, _I, _P, _I,
                                   _P, _I, _P, _P, _P], _I),
    'bmnas_bn_relu_ln_bwd_pair': ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, Dropout, _PP, _PP, _I,
                                   _U32, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I64, _P, _P], _I),
    'bmnas_bn_relu_ln_bwd': ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, Dropout, _P], _I),
    'bmnas_bn_bwd_apply': ([_P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    'bmnas_linear_fwd': ([_P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    'bmnas_linear_bwd': ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P], _I),
    'bmnas_bce_logits': ([_P, _P, _P, _P, _I, _P], _I),
    'bmnas_cross_entropy': ([_P, _P, _P, _P, _P, _I, _I, _P], _I),
    'bmnas_adam_chunk_elems': ([], _I),
    'bmnas_adam_multi': ([_P, _P, _I, _P, _P], _I),
    'bmnas_copy_batch_max': ([], _I),
    'bmnas_copy_blob_max': ([], _I),
    'bmnas_copy_batch': ([_PP, _PP, C.POINTER(C.c_longlong), _I, _P, _P, _I, _P, C.c_ulonglong, _P], _I),
    'bmnas_arch_softmax_fwd': ([_P, _P, _I, _I, _P], _I),
    'bmnas_arch_softmax_bwd': ([_P, _P, _P, _I, _I, _P], _I),
    'bmnas_backward_epilogue': ([_I, _PP, _PP, C.POINTER(_PP), C.POINTER(C.c_int), _PP, _PP, _PP, _PP, _PP,
                                 _PP, _I, C.POINTER(C.c_int), _I, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 _PP, _PP, _PP, C.POINTER(C.c_int), C.POINTER(C.c_int), _I, _I, _I64,
                                 _I, _PP, _PP, C.POINTER(C.c_int), C.POINTER(C.c_int64), _P], _I),
    'bmnas_head_chunks': ([_I], _I),
    'bmnas_head_fwd': ([_PP, _PP, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    'bmnas_debug_stamps': ([_P, _I], _I),
    'bmnas_debug_stamps_head': ([_P, _I], _I),
    'bmnas_debug_stamps_conv': ([_P, _I], _I),
    'bmnas_lazy_ln_ok': ([_I, _I], _I),
    'bmnas_lazy_ln_parts': ([_I, _I], _I),
    'bmnas_node_mix_pre_fwd': ([_P, _P, _P, _P, _P, BnFin, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, Dropout, Dropout,
                                _P], _I),
    'bmnas_mixsum_pair_fwd_lazy': ([_PP, _I, _P, _I, _P, _I, C.POINTER(LazyLn), _P, _P, _P, _P, _I, _I, _I, _P], _I),
    'bmnas_mixsum_pair_bwd_lazy': ([_PP, _PP, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I64, _U32,
                                    C.POINTER(LazyLn), _PP, C.POINTER(C.c_int), _I, _P, _I, _I, _I, _P], _I),
    'bmnas_mixsum_pair_bwd_x': ([_PP, _PP, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I64, _U32, _PP, _PP, _I,
                                 _I64, _P], _I),
    'bmnas_head_fwd_lazy': ([_PP, _PP, _I, _I, C.POINTER(LazyLn), _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P], _I),
    'bmnas_head_fwd_part_floats': ([_I, _I, _I, _I, _I], _I),
    'bmnas_node_mix_lnp_bwd_rows': ([_I], _I),
    'bmnas_conv1x1_set_deterministic': ([_I], _I),
    'bmnas_ln_set_deterministic': ([_I], _I),
    'bmnas_head_bwd_lazy': ([C.POINTER(LazyLn), _PP, _PP, _I, _U32, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P,
                             _I, _I, _I, _I, _P, _I64, _P, _P], _I),
    'bmnas_node_mix_lnp_bwd': ([_P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I64,
                                _P, _P, _U32, _P, _P, _I, _I, _I, Dropout, Dropout, _P, _P], _I),
    'bmnas_head_bwd': ([_PP, _PP, _PP, _I, _U32, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I,
                        _P, _I64, _P], _I),
    'bmnas_sum_chunks': ([_P, _P, _I, _I64, _P], _I),
    'bmnas_comm_available': ([], _I),
    'bmnas_comm_unique_id_bytes': ([], _I),
    'bmnas_comm_get_unique_id': ([_P], _I),
    'bmnas_comm_init_rank': ([C.POINTER(C.c_void_p), _I, _I, _P], _I),
    'bmnas_comm_destroy': ([_P], _I),
    'bmnas_allreduce_f32': ([_P, _I64, _I, _P, _P], _I),
    'bmnas_comm_info': ([_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)], _I),
    'bmnas_cell_prologue': ([_PP, _PP, C.POINTER(C.c_int), C.POINTER(C.c_int), _I, _PP, _PP, _I, _I, _I, _P, _P,
                             _P, _I64, _P], _I),
    'bmnas_cell_prologue_pair': ([_PP, _PP, C.POINTER(C.c_int), C.POINTER(C.c_int), _I, _PP, _PP, _I, _I, _I, _P,
                                  _P, _P, _I64, _PP, _I, _P, _P, _P, _P, _I64, _P], _I),
    'bmnas_arch_softmax_multi': ([_PP, _PP, _PP, C.POINTER(C.c_int), C.POINTER(C.c_int), _I, _I, _I, _I64,
                                 _P], _I),
}

_lib = None


def load():
    """Load the shared library (once).  Raises BmnasError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BmnasError(f'{LIB_PATH} is missing: build it with `python -m bmnas.build` '
                         '(hipcc --offload-arch=gfx950); there is no fallback path')
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def _check(rc, name):
    if rc != 0:
        kind = {-1: 'bad argument', -2: 'unsupported shape', -3: 'limit exceeded',
                -4: 'librccl could not be loaded'}.get(rc, f'ncclResult {rc - 1000}' if rc > 1000 else f'hipError {rc}')
        raise BmnasError(f'{name} failed: {kind} (rc={rc})')


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), \
        (t.device, t.dtype, t.is_contiguous())
    return t.data_ptr()


def _opt(t):
    """data_ptr of a (possibly strided) tensor, or NULL."""
    return None if t is None else t.data_ptr()


def _ptrs(ts):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = _ptr(t)
    return arr


def version():
    return load().bmnas_version()


def make_dropout(p, seed, offset, step_ptr=None):
    """thr = p * 2^32; p == 0 (or eval) -> NO_DROP.  step_ptr: device address of a uint64
    counter added to the offset at run time (None: host-side offsets only)."""
    if p <= 0.0:
        return NO_DROP
    if p >= 1.0:
        raise ValueError('dropout p must be < 1')
    thr = int(p * 4294967296.0)
    if thr <= 0:
        return NO_DROP
    return Dropout(min(thr, 0xFFFFFFFF), 1.0 / (1.0 - p), seed & 0xFFFFFFFFFFFFFFFF,
                   offset & 0xFFFFFFFFFFFFFFFF, step_ptr)


def dropout_mask(drop, numel, device, step_value=None):
    """The multipliers (0 or 1/(1-p)) the kernels apply at the dropout site `drop` to the `numel` elements of
    its output, as a flat fp32 tensor (bmnas_dropout_mask).  step_value: use this value of the device step
    counter instead of reading it (a site of a captured step, asked about after later replays)."""
    out = torch.empty(numel, device=device, dtype=torch.float32)
    if drop.thr == 0:
        return out.fill_(1.0)
    if step_value is not None:
        drop = Dropout(drop.thr, drop.scale, drop.seed, (drop.offset + int(step_value)) & 0xFFFFFFFFFFFFFFFF, None)
    _check(load().bmnas_dropout_mask(drop, numel, out.data_ptr(), _stream()), 'dropout_mask')
    return out


# ------------------------------------------------------------------------- wrappers
def mixsum_fwd(xs, w, w_stride, out):
    """w: tensor whose data_ptr() is the weight of edge 0; edge j at +j*w_stride floats."""
    _check(load().bmnas_mixsum_fwd(_ptrs(xs), len(xs), w.data_ptr(), w_stride, _ptr(out),
                                   out.numel(), _stream()), 'mixsum_fwd')


def mixsum_bwd(xs, dxs, w, w_stride, g, dw, acc_mask, dw_shards=1, dw_shard_stride=0, g2=None):
    _check(load().bmnas_mixsum_bwd(_ptrs(xs), _ptrs(dxs), len(xs), w.data_ptr(), w_stride, _ptr(g), _ptr(g2),
                                   None if dw is None else dw.data_ptr(), dw_shards, dw_shard_stride,
                                   acc_mask, g.numel(), _stream()), 'mixsum_bwd')


def mixsum_pair_fwd(xs, w, w_stride, w2, w2_stride, out, out2):
    _check(load().bmnas_mixsum_pair_fwd(_ptrs(xs), len(xs), w.data_ptr(), w_stride, w2.data_ptr(), w2_stride,
                                        _ptr(out), _ptr(out2), out.numel(), _stream()), 'mixsum_pair_fwd')


def mixsum_pair_bwd(xs, dxs, w, w_stride, w2, w2_stride, h, gh, gz, dw, dw2, acc_mask, dw_shards=1,
                    dw_shard_stride=0, gz2=None):
    _check(load().bmnas_mixsum_pair_bwd(_ptrs(xs), _ptrs(dxs), len(xs), w.data_ptr(), w_stride,
                                        w2.data_ptr(), w2_stride, _ptr(h), _ptr(gh), _ptr(gz), _ptr(gz2),
                                        _opt(dw), _opt(dw2), dw_shards, dw_shard_stride, acc_mask,
                                        gz.numel(), _stream()), 'mixsum_pair_bwd')


def cat_ln_fwd(srcs, resid, ln_w, ln_b, out, stats, b, Cc, L, relu, out_sums=None):
    """out_sums: optional (b, 2) buffer receiving each sample's (sum, sum of squares) of `out`."""
    _check(load().bmnas_cat_ln_fwd(_ptrs(srcs), len(srcs), _ptr(resid), _ptr(ln_w), _ptr(ln_b),
                                   _ptr(out), _ptr(stats), b, Cc, L, int(relu), _ptr(out_sums), _stream()),
           'cat_ln_fwd')


def cat_ln_bwd(g, srcs, resid, ln_w, ln_b, stats, dsrcs, dresid, acc_mask, dln_w, dln_b, b, Cc, L, relu,
               scrub=None):
    """scrub: a flat fp32 tensor (numel % 4 == 0) that the launch also zero-fills."""
    _check(load().bmnas_cat_ln_bwd(_ptr(g), _ptrs(srcs), len(srcs), _ptr(resid), _ptr(ln_w), _ptr(ln_b),
                                   _ptr(stats), _ptrs(dsrcs), _ptr(dresid), acc_mask, _ptr(dln_w),
                                   _ptr(dln_b), b, Cc, L, int(relu), _ptr(scrub),
                                   0 if scrub is None else scrub.numel(), _stream()), 'cat_ln_bwd')


def ln_affine_bwd(g, gscale, srcs, resid, ln_w, ln_b, stats, dln_w, dln_b, b, Cc, L, relu, prenorm):
    _check(load().bmnas_ln_affine_bwd(_ptr(g), None if gscale is None else gscale.data_ptr(), _ptrs(srcs),
                                      len(srcs), _ptr(resid), _ptr(ln_w), _ptr(ln_b), _ptr(stats),
                                      _ptr(dln_w), _ptr(dln_b), b, Cc, L, int(relu), int(prenorm),
                                      _stream()), 'ln_affine_bwd')


def ln_affine_bwd_multi(probs, b, L):
    """probs: list of dicts(g, gscale, srcs, resid, ln_w, ln_b, stats, dln_w, dln_b, C, relu, prenorm)."""
    n = len(probs)

    def parr(key):
        return (C.c_void_p * n)(*[None if p[key] is None else p[key].data_ptr() for p in probs])

    src_arrays = [_ptrs(p['srcs']) for p in probs]
    srcs = (_PP * n)(*[C.cast(a, _PP) for a in src_arrays])
    ints = lambda key: (C.c_int * n)(*[int(p[key]) for p in probs])
    n_src = (C.c_int * n)(*[len(p['srcs']) for p in probs])
    _check(load().bmnas_ln_affine_bwd_multi(n, parr('g'), parr('gscale'), srcs, n_src, parr('resid'),
                                            parr('ln_w'), parr('ln_b'), parr('stats'), parr('dln_w'),
                                            parr('dln_b'), b, ints('C'), L, ints('relu'), ints('prenorm'),
                                            _stream()), 'ln_affine_bwd_multi')


def backward_epilogue(probs, b, L, ws, dws, outs, n_shards, shard_stride, sums=()):
    """ln_affine_bwd_multi(probs) + arch_softmax_multi(ws, dws, outs, backward) in one launch.
    sums: up to two (part, out, n_chunk) — out[e] = sum_c part[c, e], the head's partials."""
    n = len(probs)

    def parr(key):
        return (C.c_void_p * n)(*[None if p[key] is None else p[key].data_ptr() for p in probs])

    src_arrays = [_ptrs(p['srcs']) for p in probs]
    srcs = (_PP * n)(*[C.cast(a, _PP) for a in src_arrays])
    ints = lambda key: (C.c_int * n)(*[int(p[key]) for p in probs])
    n_src = (C.c_int * n)(*[len(p['srcs']) for p in probs])
    na = len(ws)
    rows = (C.c_int * na)(*[t.shape[0] for t in ws])
    cols = (C.c_int * na)(*[t.shape[1] for t in ws])
    pw = (C.c_void_p * na)(*[t.data_ptr() for t in ws])
    pd = (C.c_void_p * na)(*[t.data_ptr() for t in dws])
    po = (C.c_void_p * na)(*[t.data_ptr() for t in outs])
    ns = len(sums)
    sp = (C.c_void_p * max(ns, 1))(*[t[0].data_ptr() for t in sums])
    so = (C.c_void_p * max(ns, 1))(*[t[1].data_ptr() for t in sums])
    sc = (C.c_int * max(ns, 1))(*[int(t[2]) for t in sums])
    sn = (C.c_int64 * max(ns, 1))(*[t[1].numel() for t in sums])
    for part, out, n_chunk in sums:
        assert part.numel() == n_chunk * out.numel() and out.numel() % 4 == 0
    _check(load().bmnas_backward_epilogue(n, parr('g'), parr('gscale'), srcs, n_src, parr('resid'),
                                          parr('ln_w'), parr('ln_b'), parr('stats'), parr('dln_w'),
                                          parr('dln_b'), b, ints('C'), L, ints('relu'), ints('prenorm'),
                                          pw, pd, po, rows, cols, na, n_shards, shard_stride, ns, sp, so, sc,
                                          sn, _stream()),
           'backward_epilogue')


def head_chunks(b):
    return load().bmnas_head_chunks(b)


def head_fwd(srcs, sums, ln_w, ln_b, W, bias, hb, stats, b, Cc, L, O):
    """K7 + central_classifier forward (csrc/head.hip); hb: zero-filled (3, b, O) = logits | A | B."""
    _check(load().bmnas_head_fwd(_ptrs(srcs), _ptrs(sums), len(srcs), _ptr(ln_w), _ptr(ln_b), _ptr(W),
                                 _ptr(bias), _ptr(hb), _ptr(stats), b, Cc, L, O, _stream()), 'head_fwd')


def head_bwd(srcs, sums, dsrcs, acc_mask, ln_w, ln_b, W, hb, stats, mode, g, gscale, labels, loss, part,
             b, Cc, L, O, scrub=None):
    """mode 0: g = dlogits; 1: BCEWithLogits(mean) vs float labels; 2: CrossEntropy(mean) vs int64 labels."""
    _check(load().bmnas_head_bwd(_ptrs(srcs), _ptrs(sums), _ptrs(dsrcs), len(srcs), acc_mask, _ptr(ln_w),
                                 _ptr(ln_b), _ptr(W), _ptr(hb), _ptr(stats), mode, _ptr(g),
                                 None if gscale is None else gscale.data_ptr(),
                                 None if labels is None else labels.data_ptr(), _ptr(loss), _ptr(part),
                                 b, Cc, L, O, _ptr(scrub),
                                 0 if scrub is None else scrub.numel(), _stream()), 'head_bwd')


# ------------------------------------------------- the step node's LayerNorm applied by its consumers (csrc/lazyln.hip)
def lazy_ln_ok(Cc, L):
    return bool(load().bmnas_lazy_ln_ok(Cc, L))


def lazy_ln_parts(Cc, L):
    n = load().bmnas_lazy_ln_parts(Cc, L)
    if n < 1:
        _check(n, 'lazy_ln_parts')
    return n


def make_lazy(pre, rec, prm, ln_w, ln_b, stats):
    return LazyLn(_ptr(pre), _ptr(rec), _ptr(prm), _ptr(ln_w), _ptr(ln_b), _ptr(stats))


def node_mix_pre_fwd(x, y, p1, U, chan, gamma, resid, ln_w, ln_b, pre, rec, prm, b, Cc, L, dglu, dfc, fin=NO_FIN):
    _check(load().bmnas_node_mix_pre_fwd(_ptr(x), _ptr(y), _ptr(p1), _ptr(U), _ptr(chan), fin, gamma.data_ptr(),
                                         _ptr(resid), _ptr(ln_w), _ptr(ln_b), _ptr(pre), _ptr(rec), _ptr(prm),
                                         b, Cc, L, dglu, dfc, _stream()), 'node_mix_pre_fwd')


def mixsum_pair_fwd_lazy(xs, w, w_stride, w2, w2_stride, lazy, last_out, last_sums, out, out2, b, Cc, L):
    """xs: the plain inputs; the lazy node output is the LAST input (weight w[len(xs) * w_stride])."""
    _check(load().bmnas_mixsum_pair_fwd_lazy(_ptrs(xs), len(xs), w.data_ptr(), w_stride, w2.data_ptr(), w2_stride,
                                             C.byref(lazy), _ptr(last_out), _ptr(last_sums), _ptr(out), _ptr(out2),
                                             b, Cc, L, _stream()), 'mixsum_pair_fwd_lazy')


def mixsum_pair_bwd_lazy(xs, dxs, w, w_stride, w2, w2_stride, h, gh, gz, dw, dw2, acc_mask, lazies, lnparts, strides,
                         b, Cc, L, dw_shards=1, dw_shard_stride=0, gz2=None, g_full=None):
    """The last len(lazies) inputs of xs are lazy-LayerNorm node outputs; lnparts[t] (views into (b, strides[t], 2)
    buffers, starting at this consumer's first pair) receive the partials."""
    n = len(lazies)
    arr = (LazyLn * n)(*lazies)
    st = (C.c_int * n)(*strides)
    _check(load().bmnas_mixsum_pair_bwd_lazy(_ptrs(xs), _ptrs(dxs), len(xs), w.data_ptr(), w_stride, w2.data_ptr(),
                                             w2_stride, _ptr(h), _ptr(gh), _ptr(gz), _ptr(gz2), _opt(dw),
                                             _opt(dw2), dw_shards, dw_shard_stride, acc_mask, arr,
                                             (C.c_void_p * n)(*[t.data_ptr() for t in lnparts]), st, n,
                                             _ptr(g_full), b, Cc, L, _stream()),
           'mixsum_pair_bwd_lazy')


def mixsum_pair_bwd_x(xs, dxs, w, w_stride, w2, w2_stride, h, gh, gz, dw, dw2, acc_mask, g_more, w_more, dw_shards=1,
                      dw_shard_stride=0, gz2=None):
    """g_more[t]: the stored G of a later cell step, w_more[t]: that step's softmaxed edge-weight column (first edge)."""
    n = len(g_more)
    _check(load().bmnas_mixsum_pair_bwd_x(_ptrs(xs), _ptrs(dxs), len(xs), w.data_ptr(), w_stride, w2.data_ptr(),
                                          w2_stride, _ptr(h), _ptr(gh), _ptr(gz), _ptr(gz2), _opt(dw),
                                          _opt(dw2), dw_shards, dw_shard_stride, acc_mask, _ptrs(g_more),
                                          (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in w_more]), n, gz.numel(),
                                          _stream()), 'mixsum_pair_bwd_x')


def head_fwd_lazy(srcs, sums, lazy_q, lazy, ln_w, ln_b, W, bias, hb, stats, b, Cc, L, O, hb_part=None):
    """bmnas_head_fwd whose source lazy_q is srcs[lazy_q] = the un-normalised `pre` described by `lazy`.
    hb_part (deterministic mode): head_fwd_part_floats(...) floats; hb then needs no zero-fill."""
    _check(load().bmnas_head_fwd_lazy(_ptrs(srcs), _ptrs(sums), len(srcs), lazy_q, C.byref(lazy), _ptr(ln_w),
                                      _ptr(ln_b), _ptr(W), _ptr(bias), _ptr(hb), _ptr(stats), b, Cc, L, O,
                                      _ptr(hb_part), _stream()), 'head_fwd_lazy')


def head_fwd_part_floats(b, Cc, L, n_src, O):
    n = load().bmnas_head_fwd_part_floats(b, Cc, L, n_src, O)
    if n < 0:
        _check(n, 'head_fwd_part_floats')
    return n


def node_mix_lnp_bwd_rows(b):
    return load().bmnas_node_mix_lnp_bwd_rows(b)


def set_deterministic(on):
    """The two host-side choices of the deterministic mode that live in the library (bmnas.cell.DETERMINISTIC sets
    them): weight-gradient tiles without batch splits, LayerNorm-affine reductions in one chunk."""
    _check(load().bmnas_conv1x1_set_deterministic(int(bool(on))), 'conv1x1_set_deterministic')
    _check(load().bmnas_ln_set_deterministic(int(bool(on))), 'ln_set_deterministic')


def head_bwd_lazy(lazies, lnparts, dsrcs, acc_mask, ln_w, ln_b, W, hb, stats, mode, g, gscale, labels, loss, part,
                  b, Cc, L, O, scrub=None, loss_part=None):
    n = len(lazies)
    arr = (LazyLn * n)(*lazies)
    _check(load().bmnas_head_bwd_lazy(arr, _ptrs(lnparts), _ptrs(dsrcs), n, acc_mask, _ptr(ln_w), _ptr(ln_b),
                                      _ptr(W), _ptr(hb), _ptr(stats), mode, _ptr(g),
                                      None if gscale is None else gscale.data_ptr(),
                                      None if labels is None else labels.data_ptr(), _ptr(loss), _ptr(part),
                                      b, Cc, L, O, _ptr(scrub), 0 if scrub is None else scrub.numel(), _ptr(loss_part),
                                      _stream()),
           'head_bwd_lazy')


def node_mix_lnp_bwd(gy, pre, ln_w, stats, lnp0, lnp1, g_in, dresid, acc_resid, x, y, p1, U, chan, gamma, dgamma, dx,
                     dy, acc_mask, dV, bn_grad, b, Cc, L, dglu, dfc, dg_shards=1, dg_stride=0, bn_part=None):
    """lnp0 / lnp1: (b, n, 2) partial sums of the LayerNorm backward (None: absent)."""
    n0 = 0 if lnp0 is None else lnp0.numel() // (2 * b)
    n1 = 0 if lnp1 is None else lnp1.numel() // (2 * b)
    _check(load().bmnas_node_mix_lnp_bwd(_ptr(gy), _ptr(pre), _ptr(ln_w), _ptr(stats), _ptr(lnp0), n0, _ptr(lnp1),
                                         n1, _ptr(g_in), _ptr(dresid), acc_resid, _ptr(x), _ptr(y), _ptr(p1),
                                         _ptr(U), _ptr(chan), gamma.data_ptr(),
                                         None if dgamma is None else dgamma.data_ptr(), dg_shards, dg_stride,
                                         _ptr(dx), _ptr(dy), acc_mask, _ptr(dV), _ptr(bn_grad), b, Cc, L, dglu, dfc,
                                         _ptr(bn_part), _stream()), 'node_mix_lnp_bwd')


def adaptive_maxpool_group(xs, dims, outs, idxs, b):
    """Forward of the grouped AdaptiveMaxPool2d: xs[i] contiguous (b, C_i, ...) viewed as (b, C_i, H_i, W_i) with
    dims[i] = (C, H, W, oh, ow); outs[i] (b, C, oh * ow); idxs[i] int32 like outs[i], or None."""
    n = len(xs)
    probs = (PoolProb * n)()
    for i in range(n):
        Cc, H, W, oh, ow = dims[i]
        probs[i] = PoolProb(_ptr(xs[i]), _ptr(outs[i]), None if idxs[i] is None else idxs[i].data_ptr(), None, None,
                            Cc, H, W, oh, ow)
    _check(load().bmnas_adaptive_maxpool_fwd_group(probs, n, b, _stream()), 'adaptive_maxpool_fwd_group')


def adaptive_maxpool_group_bwd(gs, idxs, dxs, dims, b):
    n = len(gs)
    probs = (PoolProb * n)()
    for i in range(n):
        Cc, H, W, oh, ow = dims[i]
        probs[i] = PoolProb(None, None, idxs[i].data_ptr(), _ptr(gs[i]), _ptr(dxs[i]), Cc, H, W, oh, ow)
    _check(load().bmnas_adaptive_maxpool_bwd_group(probs, n, b, _stream()), 'adaptive_maxpool_bwd_group')


def conv1x1_group_ok(c_ins, b, L, M):
    """Whether the N convs (C_in_i -> M on (b, C_in_i, L)) can go through the grouped entry points."""
    if not 1 <= len(c_ins) <= MAX_GROUP:
        return False
    arr = (C.c_int * len(c_ins))(*[int(c) for c in c_ins])
    return bool(load().bmnas_conv1x1_group_ok(len(c_ins), arr, b, L, M))


def conv1x1_fwd_group(srcs, Ws, biases, Us, stats, stat_shards, b, L, M):
    """U_i = W_i x_i + bias_i for every layer of the group in ONE launch; stats[i]: zero-filled BatchNorm sums
    (stat_shards copies) or None with stat_shards 0."""
    n = len(srcs)
    probs = (ConvFwdProb * n)()
    for i in range(n):
        W = Ws[i]
        probs[i] = ConvFwdProb(_ptr(srcs[i]), _ptr(W), _ptr(biases[i]), _ptr(Us[i]),
                               None if stats is None or stats[i] is None else _ptr(stats[i]), srcs[i].shape[1],
                               W.shape[1])
    _check(load().bmnas_conv1x1_fwd_group(probs, n, stat_shards, b, L, M, _stream()), 'conv1x1_fwd_group')


def bn_relu_fwd_group(Us, chans, outs, fins, drops, b, M, L):
    n = len(Us)
    probs = (BnReluFwdProb * n)()
    for i in range(n):
        probs[i] = BnReluFwdProb(_ptr(Us[i]), _ptr(chans[i]), _ptr(outs[i]), fins[i], drops[i])
    _check(load().bmnas_bn_relu_fwd_group(probs, n, b, M, L, _stream()), 'bn_relu_fwd_group')


def bn_relu_bwd_group(gs, Us, chans, dVs, bn_grads, drops, b, M, L):
    n = len(gs)
    probs = (BnReluBwdProb * n)()
    for i in range(n):
        probs[i] = BnReluBwdProb(_ptr(gs[i]), _ptr(Us[i]), _ptr(chans[i]), _ptr(dVs[i]), _ptr(bn_grads[i]), drops[i])
    _check(load().bmnas_bn_relu_bwd_group(probs, n, b, M, L, _stream()), 'bn_relu_bwd_group')


def conv1x1_bwd_group(dVs, Ws, srcs, dsrcs, dWs, dbiases, bn_Us, bn_chans, bn_grads, training, b, L, M):
    """Weight / bias gradients (+=) and, where dsrcs[i] is not None, the input gradient of every layer of the
    group in ONE launch, the BatchNorm input gradient applied on the fly (dVs = gradients w.r.t. the BatchNorm
    outputs, bn_grads already reduced)."""
    n = len(dVs)
    probs = (ConvBwdProb * n)()
    for i in range(n):
        probs[i] = ConvBwdProb(_ptr(dVs[i]), _ptr(Ws[i]), _ptr(srcs[i]), _ptr(dsrcs[i]), _ptr(dWs[i]),
                               _ptr(dbiases[i]), _ptr(bn_Us[i]), _ptr(bn_chans[i]), _ptr(bn_grads[i]),
                               srcs[i].shape[1], Ws[i].shape[1], dWs[i].shape[1], 0)
    _check(load().bmnas_conv1x1_bwd_group(probs, n, int(training), b, L, M, _stream()), 'conv1x1_bwd_group')


def comm_available():
    return bool(load().bmnas_comm_available())


def comm_get_unique_id():
    """-> bytes (rank 0 calls this and ships them to every rank)."""
    buf = C.create_string_buffer(load().bmnas_comm_unique_id_bytes())
    _check(load().bmnas_comm_get_unique_id(buf), 'comm_get_unique_id')
    return buf.raw


def comm_init_rank(world, rank, uid):
    """Collective over all ranks; -> opaque communicator handle."""
    comm = C.c_void_p()
    _check(load().bmnas_comm_init_rank(C.byref(comm), world, rank, C.create_string_buffer(uid, len(uid))),
           'comm_init_rank')
    return comm


def comm_destroy(comm):
    _check(load().bmnas_comm_destroy(comm), 'comm_destroy')


def comm_info(comm):
    """-> dict(ranks, rank, hip_device, rccl_version) as the communicator reports them (ncclCommCount, ...)."""
    v = [C.c_int(-1) for _ in range(4)]
    _check(load().bmnas_comm_info(comm, *[C.byref(x) for x in v]), 'comm_info')
    return dict(ranks=v[0].value, rank=v[1].value, hip_device=v[2].value, rccl_version=v[3].value)


def allreduce_f32(buf, comm, average=False):
    """In place, asynchronous on torch's current stream (capturable into a hipGraph)."""
    _check(load().bmnas_allreduce_f32(_ptr(buf), buf.numel(), int(average), comm, _stream()), 'allreduce_f32')


def sum_chunks(part, out, n_chunk):
    _check(load().bmnas_sum_chunks(_ptr(part), _ptr(out), n_chunk, out.numel(), _stream()), 'sum_chunks')


def sdpa_ln_fwd(x, y, ln_w, ln_b, out, xhat, stats, b, Cc, L, drop):
    _check(load().bmnas_sdpa_ln_fwd(_ptr(x), _ptr(y), _ptr(ln_w), _ptr(ln_b), _ptr(out), _ptr(xhat),
                                    _ptr(stats), b, Cc, L, drop, _stream()), 'sdpa_ln_fwd')


def sdpa_ln_bwd(g, gscale, x, y, ln_w, xhat, stats, dx, dy, acc_mask, b, Cc, L, drop):
    _check(load().bmnas_sdpa_ln_bwd(_ptr(g), None if gscale is None else gscale.data_ptr(), _ptr(x),
                                    _ptr(y), _ptr(ln_w), _ptr(xhat), _ptr(stats), _ptr(dx), _ptr(dy),
                                    acc_mask, b, Cc, L, drop, _stream()), 'sdpa_ln_bwd')


def conv1x1_num_partials(b, L):
    n = load().bmnas_conv1x1_num_partials(b, L)
    if n < 0:
        _check(n, 'conv1x1_num_partials')
    return n


def conv1x1_fwd(srcs, C_src, W, ldw, bias, U, part, b, L, M, fold=0, stat_shards=0):
    """stat_shards > 0: `part` is a zero-filled (stat_shards, M, 2) buffer of running batch sums."""
    _check(load().bmnas_conv1x1_fwd(_ptrs(srcs), len(srcs), C_src, W.data_ptr(), ldw, fold, _ptr(bias),
                                    _ptr(U), _ptr(part), stat_shards, b, L, M, _stream()), 'conv1x1_fwd')


def conv1x1_bwd_data(dU, W, ldw, dsrcs, C_src, acc_mask, b, L, M, fold=0):
    _check(load().bmnas_conv1x1_bwd_data(_ptr(dU), W.data_ptr(), ldw, fold, _ptrs(dsrcs), len(dsrcs),
                                         C_src, acc_mask, b, L, M, _stream()), 'conv1x1_bwd_data')


def conv1x1_fwd_sdpa(srcs, C_src, W, ldw, bias, U, part, b, L, M, fold, x, y, ln_w, ln_b, out, xhat, stats, Cc,
                     drop, stat_shards=0):
    _check(load().bmnas_conv1x1_fwd_sdpa(_ptrs(srcs), len(srcs), C_src, W.data_ptr(), ldw, fold, _ptr(bias),
                                         _ptr(U), _ptr(part), stat_shards, b, L, M, _ptr(x), _ptr(y), _ptr(ln_w),
                                         _ptr(ln_b), _ptr(out), _ptr(xhat), _ptr(stats), Cc, drop, _stream()),
           'conv1x1_fwd_sdpa')


def conv1x1_bwd_all_sdpa(dU, W, ldw, dsrcs, C_src, acc_mask, b, L, M, fold, wsrcs, dW, ldw_grad, dbias, dup_cols,
                         g, gscale, x, y, ln_w, xhat, stats, dx, dy, sdpa_acc_mask, Cc, drop, bn=None):
    """bn = (U, chan, bn_grad, training): dU holds dV and the launch applies the BatchNorm input
    gradient on the fly (no bn_bwd_apply launch).  dW None: no weight-gradient tiles in the launch."""
    bU, bchan, bgrad, btrain = (None, None, None, 0) if bn is None else bn
    _check(load().bmnas_conv1x1_bwd_all_sdpa(_ptr(dU), W.data_ptr(), ldw, fold, _ptrs(dsrcs), len(dsrcs),
                                             C_src, acc_mask, b, L, M, _ptrs(wsrcs),
                                             None if dW is None else dW.data_ptr(), ldw_grad,
                                             None if dbias is None else dbias.data_ptr(), dup_cols, _ptr(g),
                                             None if gscale is None else gscale.data_ptr(), _ptr(x), _ptr(y),
                                             _ptr(ln_w), _ptr(xhat), _ptr(stats), _ptr(dx), _ptr(dy),
                                             sdpa_acc_mask, Cc, drop, _ptr(bU), _ptr(bchan), _ptr(bgrad),
                                             int(btrain), _stream()), 'conv1x1_bwd_all_sdpa')


def conv1x1_bwd_all_mix_ok(b, L, M, n_src, C_src):
    return bool(load().bmnas_conv1x1_bwd_all_mix_ok(b, L, M, n_src, C_src))


def conv1x1_bwd_all(dU, W, ldw, dsrcs, C_src, acc_mask, b, L, M, fold, wsrcs, dW, ldw_grad, dbias, dup_cols,
                    bn=None, mix=None):
    """bn_bwd_apply (bn = (U, chan, bn_grad, training)) + data gradient + weight gradient of a conv
    without an attention branch; one launch at small grids.
    mix = (q, U, chan, x, p1, gamma, dgamma, dg_shards, dg_stride, dx, acc_dx, dV, bn_grad, dglu, dfc): the
    NodeMixedOp backward behind source q as the epilogue of its data-gradient tiles (bmnas_conv1x1_bwd_all_mix)."""
    bU, bchan, bgrad, btrain = (None, None, None, 0) if bn is None else bn
    if mix is None:
        _check(load().bmnas_conv1x1_bwd_all(_ptr(dU), W.data_ptr(), ldw, fold, _ptrs(dsrcs), len(dsrcs), C_src,
                                            acc_mask, b, L, M, _ptrs(wsrcs), dW.data_ptr(), ldw_grad,
                                            None if dbias is None else dbias.data_ptr(), dup_cols, _ptr(bU),
                                            _ptr(bchan), _ptr(bgrad), int(btrain), _stream()), 'conv1x1_bwd_all')
        return
    q, mU, mchan, mx, mp1, mgamma, mdgamma, mshards, mstride, mdx, macc, mdV, mbn, dglu, dfc = mix
    ep = MixEp(_ptr(mU), _ptr(mchan), _ptr(mx), _ptr(mp1), mgamma.data_ptr(),
               None if mdgamma is None else mdgamma.data_ptr(), mshards, mstride, _ptr(mdx), int(macc), _ptr(mdV),
               _ptr(mbn), q, dglu, dfc)
    _check(load().bmnas_conv1x1_bwd_all_mix(_ptr(dU), W.data_ptr(), ldw, fold, _ptrs(dsrcs), len(dsrcs), C_src,
                                            acc_mask, b, L, M, _ptrs(wsrcs), dW.data_ptr(), ldw_grad,
                                            None if dbias is None else dbias.data_ptr(), dup_cols, _ptr(bU),
                                            _ptr(bchan), _ptr(bgrad), int(btrain), C.byref(ep), _stream()),
           'conv1x1_bwd_all_mix')


def conv1x1_bwd_weight(dU, srcs, C_src, dW, ldw, dbias, dup_cols, b, L, M):
    _check(load().bmnas_conv1x1_bwd_weight(_ptr(dU), _ptrs(srcs), len(srcs), C_src, dW.data_ptr(), ldw,
                                           None if dbias is None else dbias.data_ptr(), dup_cols, b, L,
                                           M, _stream()), 'conv1x1_bwd_weight')


def conv_family_calls(reset=False):
    """{family name: calls since the last reset} of the conv GEMM dispatch (diagnostics)."""
    lib = load()
    n = lib.bmnas_conv_family_calls(None, 0, 0)
    buf = (C.c_long * n)()
    lib.bmnas_conv_family_calls(buf, n, int(reset))
    lib.bmnas_conv_family_name.restype = C.c_char_p
    lib.bmnas_conv_family_name.argtypes = [_I]
    return {lib.bmnas_conv_family_name(i).decode(): int(buf[i]) for i in range(n)}


def probe_read(p, width, sink, row_stride=0):
    """FETCH_SIZE calibration read of the whole tensor p (diagnostics, tools/calibrate_fetch.py)."""
    _check(load().bmnas_probe_read(_ptr(p), p.numel(), width, row_stride, _ptr(sink), _stream()), 'probe_read')


def fold_weight(W, Weff, M, Cc):
    _check(load().bmnas_fold_weight(W.data_ptr(), _ptr(Weff), M, Cc, _stream()), 'fold_weight')


def bn_finalize(part, n_part, b, L, M, bn_w, bn_b, rm, rv, nbt, training, chan):
    """nbt: int64 tensor of 1..k consecutive counters (or None)."""
    _check(load().bmnas_bn_finalize(_ptr(part), n_part, b, L, M, bn_w.data_ptr(), bn_b.data_ptr(),
                                    None if rm is None else rm.data_ptr(),
                                    None if rv is None else rv.data_ptr(),
                                    None if nbt is None else nbt.data_ptr(),
                                    0 if nbt is None else nbt.numel(), int(training),
                                    chan.data_ptr(), _stream()), 'bn_finalize')


def node_mix_fwd(x, y, p1, U, chan, gamma, out, b, Cc, L, dglu, dfc, fin=NO_FIN, nxt=None):
    """nxt = (prev states, w_row0, w_stride, z_next): the next inner step's mixed sum in the same launch."""
    if nxt is None:
        _check(load().bmnas_node_mix_fwd(_ptr(x), _ptr(y), _ptr(p1), _ptr(U), _ptr(chan), fin, gamma.data_ptr(),
                                         _ptr(out), b, Cc, L, dglu, dfc, _stream()), 'node_mix_fwd')
        return
    prev, w, ws, z = nxt
    _check(load().bmnas_node_mix_fwd_next(_ptr(x), _ptr(y), _ptr(p1), _ptr(U), _ptr(chan), fin, gamma.data_ptr(),
                                          _ptr(out), b, Cc, L, dglu, dfc, _ptrs(prev), len(prev), w.data_ptr(),
                                          ws, _ptr(z), _stream()), 'node_mix_fwd_next')


def node_mix_conv_fwd_ok(b, Cc, L, n_src):
    return bool(load().bmnas_node_mix_conv_fwd_ok(b, Cc, L, n_src))


def node_mix_conv_fwd(x, y, p1, U, chan, gamma, mix_out, dglu, dfc, fin, srcs, W, ldw, bias, V, stat, stat_shards,
                      b, Cc, L):
    """The last inner step's mix + out_conv over cat(srcs, mix) in one launch (small grids)."""
    _check(load().bmnas_node_mix_conv_fwd(_ptr(x), _ptr(y), _ptr(p1), _ptr(U), _ptr(chan), fin, gamma.data_ptr(),
                                          _ptr(mix_out), dglu, dfc, _ptrs(srcs), len(srcs), _ptr(W), ldw, _ptr(bias),
                                          _ptr(V), _ptr(stat), stat_shards, b, Cc, L, _stream()),
           'node_mix_conv_fwd')


def node_mix_ln_fwd(x, y, p1, U, chan, gamma, resid, ln_w, ln_b, pre, out, stats, b, Cc, L, dglu, dfc,
                    fin=NO_FIN, out_sums=None):
    _check(load().bmnas_node_mix_ln_fwd(_ptr(x), _ptr(y), _ptr(p1), _ptr(U), _ptr(chan), fin,
                                        gamma.data_ptr(), _ptr(resid), _ptr(ln_w), _ptr(ln_b), _ptr(pre),
                                        _ptr(out), _ptr(stats), b, Cc, L, dglu, dfc, _ptr(out_sums),
                                        _stream()),
           'node_mix_ln_fwd')


def node_mix_bwd(g, x, y, p1, U, chan, gamma, dgamma, dx, dy, acc_mask, dV, bn_grad, b, Cc, L, dglu, dfc,
                 dg_shards=1, dg_stride=0, nxt=None):
    """nxt = (prev, dprev, prev_acc_mask, w_row0, w_stride, dw_row0, dw_shards, dw_stride, s, gz, gz2, g_out):
    the backward of the next inner step's mixed sum in the same launch (g: nullable, earlier contributions)."""
    if nxt is None:
        _check(load().bmnas_node_mix_bwd(_ptr(g), _ptr(x), _ptr(y), _ptr(p1), _ptr(U), _ptr(chan),
                                         gamma.data_ptr(), None if dgamma is None else dgamma.data_ptr(),
                                         dg_shards, dg_stride, _ptr(dx), _ptr(dy), acc_mask, _ptr(dV),
                                         _ptr(bn_grad), b, Cc, L, dglu, dfc, _stream()), 'node_mix_bwd')
        return
    prev, dprev, pacc, w, ws, dw, dws, dwst, s, gz, gz2, g_out = nxt
    _check(load().bmnas_node_mix_bwd_next(_ptr(g), _ptr(x), _ptr(y), _ptr(p1), _ptr(U), _ptr(chan),
                                          gamma.data_ptr(), None if dgamma is None else dgamma.data_ptr(),
                                          dg_shards, dg_stride, _ptr(dx), _ptr(dy), acc_mask, _ptr(dV),
                                          _ptr(bn_grad), b, Cc, L, dglu, dfc, _ptrs(prev), _ptrs(dprev), len(prev),
                                          pacc, w.data_ptr(), ws, dw.data_ptr(), dws, dwst, _ptr(s), _ptr(gz),
                                          _ptr(gz2), _ptr(g_out), _stream()), 'node_mix_bwd_next')


def node_mix_ln_bwd_ok(b, Cc, L):
    return bool(load().bmnas_node_mix_ln_bwd_ok(b, Cc, L))


def node_mix_ln_bwd(g, pre, ln_w, stats, g_in, dresid, acc_resid, x, y, p1, U, chan, gamma, dgamma, dx, dy,
                    acc_mask, dV, bn_grad, b, Cc, L, dglu, dfc, dg_shards=1, dg_stride=0):
    """K6 backward + K2 backward in one launch (bmnas_node_mix_ln_bwd)."""
    _check(load().bmnas_node_mix_ln_bwd(_ptr(g), _ptr(pre), _ptr(ln_w), _ptr(stats), _ptr(g_in), _ptr(dresid),
                                        int(acc_resid), _ptr(x), _ptr(y), _ptr(p1), _ptr(U), _ptr(chan),
                                        gamma.data_ptr(), None if dgamma is None else dgamma.data_ptr(),
                                        dg_shards, dg_stride, _ptr(dx), _ptr(dy), acc_mask, _ptr(dV),
                                        _ptr(bn_grad), b, Cc, L, dglu, dfc, _stream()), 'node_mix_ln_bwd')


def bn_glu_fwd(U, chan, out, b, Cc, L, drop, fin=NO_FIN):
    _check(load().bmnas_bn_glu_fwd(_ptr(U), _ptr(chan), fin, _ptr(out), b, Cc, L, drop, _stream()), 'bn_glu_fwd')


def bn_glu_bwd(g, U, chan, dV, bn_grad, b, Cc, L, drop):
    _check(load().bmnas_bn_glu_bwd(_ptr(g), _ptr(U), _ptr(chan), _ptr(dV), _ptr(bn_grad), b, Cc, L, drop,
                                   _stream()), 'bn_glu_bwd')


def bn_relu_fwd(U, chan, out, b, M, L, drop, fin=NO_FIN):
    _check(load().bmnas_bn_relu_fwd(_ptr(U), _ptr(chan), fin, _ptr(out), b, M, L, drop, _stream()),
           'bn_relu_fwd')


def bn_relu_bwd(g, U, chan, dV, bn_grad, b, M, L, drop):
    _check(load().bmnas_bn_relu_bwd(_ptr(g), _ptr(U), _ptr(chan), _ptr(dV), _ptr(bn_grad), b, M, L, drop,
                                    _stream()), 'bn_relu_bwd')


def bn_relu_ln_fwd_pair_ok(b, Cc, L, n_prev):
    return bool(load().bmnas_bn_relu_ln_fwd_pair_ok(b, Cc, L, n_prev))


def bn_relu_ln_fwd(U, chan, resid, ln_w, ln_b, o, out, stats, b, Cc, L, drop, fin=NO_FIN, out_sums=None, nxt=None):
    """nxt = (xs, w, w_stride, w2, w2_stride, h, z): the next cell step's K1 pair sum in the same launch
    (bmnas_bn_relu_ln_fwd_pair; its last input is the `out` of this call)."""
    if nxt is None:
        _check(load().bmnas_bn_relu_ln_fwd(_ptr(U), _ptr(chan), fin, _ptr(resid), _ptr(ln_w), _ptr(ln_b), _ptr(o),
                                           _ptr(out), _ptr(stats), b, Cc, L, drop, _ptr(out_sums), _stream()),
               'bn_relu_ln_fwd')
        return
    xs, w, ws, w2, w2s, h, z = nxt
    _check(load().bmnas_bn_relu_ln_fwd_pair(_ptr(U), _ptr(chan), fin, _ptr(resid), _ptr(ln_w), _ptr(ln_b), _ptr(o),
                                            _ptr(out), _ptr(stats), b, Cc, L, drop, _ptr(out_sums), _ptrs(xs),
                                            len(xs), w.data_ptr(), ws, w2.data_ptr(), w2s, _ptr(h), _ptr(z),
                                            _stream()), 'bn_relu_ln_fwd_pair')


def bn_relu_ln_bwd(g, o, resid, ln_w, stats, U, chan, dV, bn_grad, dresid, acc_resid, b, Cc, L, drop, pre=None):
    """pre = (xs, dxs, acc_mask, out, w, w_stride, w2, w2_stride, h, gh, gz, gz2, dw, dw2, dw_shards,
    dw_shard_stride, g_full): the backward of the next cell step's K1 pair sum first, in the same launch
    (bmnas_bn_relu_ln_bwd_pair; g may then be None)."""
    if pre is None:
        _check(load().bmnas_bn_relu_ln_bwd(_ptr(g), _ptr(o), _ptr(resid), _ptr(ln_w), _ptr(stats), _ptr(U),
                                           _ptr(chan), _ptr(dV), _ptr(bn_grad), _ptr(dresid), int(acc_resid), b,
                                           Cc, L, drop, _stream()), 'bn_relu_ln_bwd')
        return
    xs, dxs, acc, out, w, ws, w2, w2s, h, gh, gz, gz2, dw, dw2, shards, stride, g_full = pre
    _check(load().bmnas_bn_relu_ln_bwd_pair(_ptr(g), _ptr(o), _ptr(resid), _ptr(ln_w), _ptr(stats), _ptr(U),
                                            _ptr(chan), _ptr(dV), _ptr(bn_grad), _ptr(dresid), int(acc_resid), b, Cc,
                                            L, drop, _ptrs(xs), _ptrs(dxs), len(xs), acc, _ptr(out),
                                            w.data_ptr(), ws, w2.data_ptr(), w2s, _ptr(h), _ptr(gh), _ptr(gz),
                                            _ptr(gz2), dw.data_ptr(), dw2.data_ptr(), shards, stride,
                                            _ptr(g_full), _stream()), 'bn_relu_ln_bwd_pair')


def bn_bwd_apply(dV, U, chan, bn_grad, b, M, L, training):
    _check(load().bmnas_bn_bwd_apply(_ptr(dV), _ptr(U), _ptr(chan), _ptr(bn_grad), b, M, L,
                                     int(training), _stream()), 'bn_bwd_apply')


def arch_softmax_fwd(logits, w, rows, cols):
    _check(load().bmnas_arch_softmax_fwd(logits.data_ptr(), w.data_ptr(), rows, cols, _stream()),
           'arch_softmax_fwd')


def arch_softmax_bwd(w, dw, dlogits, rows, cols):
    _check(load().bmnas_arch_softmax_bwd(w.data_ptr(), dw.data_ptr(), dlogits.data_ptr(), rows, cols,
                                         _stream()), 'arch_softmax_bwd')


def linear_fwd(feat, W, bias, out, b, O, Kd, out_is_zero=False):
    _check(load().bmnas_linear_fwd(_ptr(feat), _ptr(W), _ptr(bias), _ptr(out), b, O, Kd, int(out_is_zero), _stream()),
           'linear_fwd')


def linear_bwd(g, gscale, feat, W, dfeat, dW, dbias, b, O, Kd):
    _check(load().bmnas_linear_bwd(_ptr(g), None if gscale is None else gscale.data_ptr(), _ptr(feat),
                                   _ptr(W), _ptr(dfeat), _ptr(dW), _ptr(dbias), b, O, Kd, _stream()),
           'linear_bwd')


def bce_logits(z, y, loss, dz):
    _check(load().bmnas_bce_logits(_ptr(z), _ptr(y), _ptr(loss), _ptr(dz), z.numel(), _stream()),
           'bce_logits')


def cross_entropy(z, label, loss, dz, row_loss, b, O):
    _check(load().bmnas_cross_entropy(_ptr(z), label.data_ptr(), _ptr(loss), _ptr(dz), _ptr(row_loss), b, O,
                                      _stream()), 'cross_entropy')


def adam_chunk_elems():
    return load().bmnas_adam_chunk_elems()


def adam_multi(table, chunks, n_chunks, hyp):
    """table: uint8 device tensor holding bmnas_adam_tensor_t[]; chunks: int32 device (n_chunks, 2);
    hyp: float32 device (rows, 8)."""
    _check(load().bmnas_adam_multi(table.data_ptr(), chunks.data_ptr(), n_chunks, hyp.data_ptr(), _stream()),
           'adam_multi')


def copy_blob_max():
    return load().bmnas_copy_blob_max()


class BatchCopier:
    """bmnas_copy_batch for a FIXED list of destinations (a captured step's static tensors), called once per batch: the
    ctypes argument arrays are built once, a call only fills in the source addresses (the host side of a replayed step is
    ~10 Python-level operations, not a list comprehension per tensor attribute).
    zero: tensors the same launch ZERO-FILLS on every call (the step's accumulation arena); advance = (int64 device
    tensor, value): a counter the launch advances by `value` on every call."""

    def __init__(self, dsts, zero=(), advance=None):
        self.dsts = list(dsts)
        self.zero = [z for z in zero if z is not None and z.numel()]
        n = len(self.dsts) + len(self.zero)
        self.cap = load().bmnas_copy_batch_max()     # tensors per launch; longer lists go out in groups of `cap`
        self.dptr = [d.data_ptr() for d in self.dsts]
        self.nbytes = [d.numel() * d.element_size() for d in self.dsts]
        self.ps = (C.c_void_p * max(n, 1))()
        self.pd = (C.c_void_p * max(n, 1))()
        self.nb = (C.c_longlong * max(n, 1))()
        self.fn = load().bmnas_copy_batch
        self.add_dst = None if advance is None else advance[0].data_ptr()
        self.add_val = 0 if advance is None else int(advance[1])
        self._keep = (self.zero, advance)

    def __call__(self, srcs, blob=None):
        """srcs[i] -> dsts[i] for every i whose source is not the destination itself; -> the (dst, src) pairs this
        launch could NOT take (another device / dtype / shape, non-contiguous): the caller copies those with torch."""
        n, slow = 0, []
        for i, s_ in enumerate(srcs):
            d = self.dsts[i]
            if s_ is d:
                continue
            if s_.device == d.device and s_.dtype == d.dtype and s_.shape == d.shape and s_.is_contiguous():
                sp = s_.data_ptr()
                if sp == self.dptr[i]:
                    continue
                self.ps[n], self.pd[n], self.nb[n] = sp, self.dptr[i], self.nbytes[i]
                n += 1
            else:
                slow.append((d, s_))
        for z in self.zero:                                  # zero-fill jobs: a NULL source
            self.ps[n], self.pd[n], self.nb[n] = None, z.data_ptr(), z.numel() * z.element_size()
            n += 1
        bd, bp, bn, keep = None, None, 0, None
        if blob is not None:
            raw = blob[1]
            bd, bn = blob[0].data_ptr(), len(raw)
            keep = C.create_string_buffer(raw, bn)
            bp = C.cast(keep, C.c_void_p)
        if n <= self.cap:
            if n or bn or self.add_dst:
                _check(self.fn(self.ps, self.pd, self.nb, n, bd, bp, bn, self.add_dst, self.add_val, _stream()),
                       'copy_batch')
            return slow
        # more tensors than one launch takes (a model with >= 15 inputs): groups of `cap`, the blob and the counter
        # advance riding in the LAST one (they must happen once)
        esz = C.sizeof(C.c_void_p)
        for o in range(0, n, self.cap):
            m = min(self.cap, n - o)
            last = o + m >= n
            _check(self.fn((C.c_void_p * m).from_buffer(self.ps, o * esz), (C.c_void_p * m).from_buffer(self.pd, o * esz),
                           (C.c_longlong * m).from_buffer(self.nb, o * C.sizeof(C.c_longlong)), m,
                           bd if last else None, bp if last else None, bn if last else 0,
                           self.add_dst if last else None, self.add_val if last else 0, _stream()), 'copy_batch')
        return slow


def copy_batch(pairs, blob=None):
    """pairs: [(dst, src)] device tensors of equal byte size, contiguous, any dtypes -> ONE launch (groups of
    bmnas_copy_batch_max() tensors).  blob = (dst device tensor, host bytes-like of <= bmnas_copy_blob_max() bytes): stored
    to dst by the same launch (it travels by value in the kernel arguments)."""
    cap = load().bmnas_copy_batch_max()
    groups = [pairs[i:i + cap] for i in range(0, len(pairs), cap)] or [[]]
    for gi, part in enumerate(groups):
        n = len(part)
        ps = (C.c_void_p * max(n, 1))(*[None if s_ is None else s_.data_ptr() for _, s_ in part])   # None: zero-fill
        pd = (C.c_void_p * max(n, 1))(*[d.data_ptr() for d, _ in part])
        nb = (C.c_longlong * max(n, 1))(*[d.numel() * d.element_size() for d, _ in part])
        bd, bp, bn, keep = None, None, 0, None
        if blob is not None and gi == len(groups) - 1:
            raw = bytes(blob[1])
            bd, bn = blob[0].data_ptr(), len(raw)
            keep = C.create_string_buffer(raw, bn)              # (alive until the call has copied it into the kernargs)
            bp = C.cast(keep, C.c_void_p)
        if n == 0 and bn == 0:
            continue
        _check(load().bmnas_copy_batch(ps, pd, nb, n, bd, bp, bn, None, 0, _stream()), 'copy_batch')


def cell_prologue(a_list, out_list, Ws, Weffs, M, Cc, step=None, scrub=None):
    """Row softmax of every arch tensor + folded conv weights of every NodeMixedOp, one launch.
    step = (counter, span): int64 device tensors; the launch also does counter += span.
    scrub: flat fp32 tensor (numel % 4 == 0) the launch also zero-fills."""
    n = len(a_list)
    rows = (C.c_int * max(n, 1))(*[t.shape[0] for t in a_list])
    cols = (C.c_int * max(n, 1))(*[t.shape[1] for t in a_list])
    pa = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in a_list])
    po = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in out_list])
    nf = len(Ws)
    pw = (C.c_void_p * max(nf, 1))(*[t.data_ptr() for t in Ws])
    pe = (C.c_void_p * max(nf, 1))(*[t.data_ptr() for t in Weffs])
    sc, sp = (None, None) if step is None else (step[0].data_ptr(), step[1].data_ptr())
    _check(load().bmnas_cell_prologue(pa, po, rows, cols, n, pw, pe, nf, M, Cc, sc, sp, _ptr(scrub),
                                      0 if scrub is None else scrub.numel(), _stream()), 'cell_prologue')


def cell_prologue_pair(a_list, out_list, Ws, Weffs, M, Cc, step, scrub, xs, alpha_logits, beta_logits, h, z):
    """cell_prologue(...) + mixsum_pair_fwd of the first step (weights from the raw logits), one launch."""
    n = len(a_list)
    rows = (C.c_int * max(n, 1))(*[t.shape[0] for t in a_list])
    cols = (C.c_int * max(n, 1))(*[t.shape[1] for t in a_list])
    pa = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in a_list])
    po = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in out_list])
    nf = len(Ws)
    pw = (C.c_void_p * max(nf, 1))(*[t.data_ptr() for t in Ws])
    pe = (C.c_void_p * max(nf, 1))(*[t.data_ptr() for t in Weffs])
    sc, sp = (None, None) if step is None else (step[0].data_ptr(), step[1].data_ptr())
    _check(load().bmnas_cell_prologue_pair(pa, po, rows, cols, n, pw, pe, nf, M, Cc, sc, sp, _ptr(scrub),
                                           0 if scrub is None else scrub.numel(), _ptrs(xs), len(xs),
                                           alpha_logits.data_ptr(), beta_logits.data_ptr(), _ptr(h), _ptr(z),
                                           h.numel(), _stream()), 'cell_prologue_pair')


def arch_softmax_multi(a_list, dw_list, out_list, backward, n_shards=1, shard_stride=0):
    """One launch for every architecture tensor (row softmax, or its backward)."""
    n = len(a_list)
    rows = (C.c_int * n)(*[t.shape[0] for t in a_list])
    cols = (C.c_int * n)(*[t.shape[1] for t in a_list])
    pa = (C.c_void_p * n)(*[t.data_ptr() for t in a_list])
    po = (C.c_void_p * n)(*[t.data_ptr() for t in out_list])
    pd = (C.c_void_p * n)(*[t.data_ptr() for t in dw_list]) if backward else None
    _check(load().bmnas_arch_softmax_multi(pa, pd, po, rows, cols, n, int(backward), n_shards,
                                           shard_stride, _stream()),
           'arch_softmax_multi')


# ----------------------------------------------------------------- optional kernel timing
# bench.py brackets selected wrappers with HIP events on the launch stream (torch's current
# stream, the one the kernels are enqueued on) to measure per-kernel durations live.
_PROF = None


def profile_begin(algo, events=True):
    """algo: {wrapper name: fn(*args) -> (bound, units)} with bound in {'hbm', 'mfma'} and
    units = algorithmic bytes / flops of that launch.  Only the named wrappers are recorded.
    events=False: no HIP events, only the ordered list of (wrapper, bound, units) — profile_end_calls()."""
    global _PROF
    _PROF = {'algo': algo, 'records': {}, 'calls': [], 'events': events}
    g = globals()
    for n in _TIMED_NAMES:
        if n in algo and n not in _PLAIN:
            _PLAIN[n] = g[n]
            g[n] = _timed(n, g[n])


def profile_end():
    """-> ({name: [(ms, bound, units), ...]}, empty_bracket_ms) (synchronises).  The second value
    is the mean elapsed time of an EMPTY start/end event pair queued in the same pass: an event
    record is a packet of its own (~3-4 us between two of them on MI355X), which is subtracted
    from every bracket by the caller."""
    global _PROF
    prof, _PROF = _PROF, None
    g = globals()
    for n, fn in list(_PLAIN.items()):
        g[n] = fn
    _PLAIN.clear()
    empties = []
    for _ in range(32):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        e.record()
        empties.append((s, e))
    torch.cuda.synchronize()
    out = {}
    for name, recs in prof['records'].items():
        out[name] = [(s.elapsed_time(e), bound, units) for s, e, bound, units in recs]
    overhead = sorted(s.elapsed_time(e) for s, e in empties)[len(empties) // 2]
    return out, overhead


def profile_end_calls():
    """-> [(wrapper, bound, units), ...] in launch order (profile_begin(algo, events=False))."""
    global _PROF
    prof, _PROF = _PROF, None
    g = globals()
    for n, fn in list(_PLAIN.items()):
        g[n] = fn
    _PLAIN.clear()
    return prof['calls']


_TIMED_NAMES = ('node_mix_pre_fwd', 'mixsum_pair_fwd_lazy', 'mixsum_pair_bwd_lazy', 'mixsum_pair_bwd_x', 'head_fwd_lazy',
                'head_bwd_lazy', 'node_mix_lnp_bwd', 'head_fwd', 'head_bwd', 'cell_prologue', 'cell_prologue_pair', 'mixsum_fwd', 'mixsum_bwd', 'mixsum_pair_fwd', 'mixsum_pair_bwd', 'cat_ln_fwd', 'cat_ln_bwd', 'ln_affine_bwd', 'ln_affine_bwd_multi', 'backward_epilogue',
                'sdpa_ln_fwd', 'sdpa_ln_bwd', 'conv1x1_fwd', 'conv1x1_bwd_data', 'conv1x1_bwd_weight',
                'conv1x1_fwd_sdpa', 'conv1x1_bwd_all_sdpa', 'conv1x1_bwd_all',
                'bn_relu_ln_fwd', 'bn_relu_ln_bwd',
                'fold_weight', 'bn_finalize', 'node_mix_fwd', 'node_mix_conv_fwd', 'node_mix_ln_fwd', 'node_mix_bwd', 'node_mix_ln_bwd', 'bn_glu_fwd', 'bn_glu_bwd',
                'bn_relu_fwd', 'bn_relu_bwd', 'bn_bwd_apply', 'arch_softmax_fwd', 'arch_softmax_bwd',
                'linear_fwd', 'linear_bwd', 'bce_logits', 'cross_entropy', 'adam_multi', 'conv1x1_fwd_group',
                'conv1x1_bwd_group', 'bn_relu_fwd_group', 'bn_relu_bwd_group')
_PLAIN = {}


def _timed(name, fn):
    def wrapper(*a, **k):
        prof = _PROF
        if prof is None or name not in prof['algo']:
            return fn(*a, **k)
        bound, units = prof['algo'][name](*a, **k)
        prof['calls'].append((name, bound, units))
        if not prof['events']:
            return fn(*a, **k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*a, **k)
        e.record()
        prof['records'].setdefault(name, []).append((s, e, bound, units))
        return r
    wrapper.__name__ = name
    wrapper.__doc__ = fn.__doc__
    return wrapper
