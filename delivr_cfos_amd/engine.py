"""HipEngine - one libdelivr_hip context bound to one MI355X, driven with torch tensors.

torch is only the device-memory container here (and, across ranks, the RCCL front end): every
computation below is a call through the C ABI (include/delivr_hip.h).  There is no CPU
fallback; constructing an engine without a visible GPU raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import PREC_BF16, PREC_BF16_ALL, PREC_F16, PREC_F32, DelivrHipError

# "bf16": bf16 operands / storage at levels 1-4, fp16 at full resolution (include/delivr_hip.h DLV_PREC_BF16);
# "bf16_all": bf16 at every level
PRECISIONS = {"fp32": PREC_F32, "f32": PREC_F32, "bf16": PREC_BF16, "fp16": PREC_F16, "f16": PREC_F16, "bf16_all": PREC_BF16_ALL}
RANGE_GUARDED = ("fp16", "f16", "bf16")  # formats with fp16 tensors: DLV_ERANGE is recoverable (range_guard.py)

# checkpoint keys of the 18 conv blocks in forward order (include/delivr_hip.h)
CONV_BLOCKS = (
    ["conv_0.conv_0", "conv_0.conv_1"]
    + [f"down_{k}.convs.conv_{j}" for k in (1, 2, 3, 4) for j in (0, 1)]
    + [f"upcat_{k}.convs.conv_{j}" for k in (4, 3, 2, 1) for j in (0, 1)]
)
DECONV_BLOCKS = [f"upcat_{k}.upsample.deconv" for k in (4, 3, 2, 1)]


def strip_state_dict(checkpoint) -> Dict[str, "np.ndarray"]:
    """Accepts what torch.load(model_weights) returns in the reference: a dict holding
    "state_dict" (inference/inference.py:222) or "model_state" (inference_nifti_load.py:215), keys
    prefixed "module." by DataParallel; or a bare state_dict."""
    sd = checkpoint
    if isinstance(sd, dict):
        for key in ("state_dict", "model_state"):
            if key in sd and isinstance(sd[key], dict):
                sd = sd[key]
                break
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        out[k] = v
    return out


class HipComm:
    """One process, N devices: dlv_comm_init_all / dlv_bcast_weights / dlv_sw_infer_sharded (the C-ABI form of the
    DataParallel replacement; the multi-process form lives in parallel.py).  `engines[r]` drives rank r's context."""

    def __init__(self, devices: Sequence[int]):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("delivr_cfos_amd needs an MI355X: there is no CPU fallback for the HIP path")
        self.torch = torch
        self.lib = _lib.load()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        rc = self.lib.dlv_comm_init_all(len(self.devices), arr, C.byref(h))
        if rc != 0:
            why = self.lib.dlv_comm_last_error(None)  # (NULL: the reason the last init of this thread failed)
            raise DelivrHipError(rc, "dlv_comm_init_all failed: " + (why.decode() if why else "several devices need librccl.so"))
        self.handle = h
        self.engines = [HipEngine(d, _ctx=C.c_void_p(self.lib.dlv_comm_ctx(h, r))) for r, d in enumerate(self.devices)]

    def close(self):
        if getattr(self, "handle", None):
            self.lib.dlv_comm_destroy(self.handle)
            self.handle = None
            for e in self.engines:
                e.ctx = None

    def _check(self, rc: int):
        if rc != 0:
            raise DelivrHipError(rc, self.lib.dlv_comm_last_error(self.handle).decode(errors="replace"))

    @property
    def uses_rccl(self) -> bool:
        return bool(self.lib.dlv_comm_uses_rccl(self.handle))

    def selftest(self, nbytes: int = 64 << 20) -> None:
        """Ring exchange + broadcast of `nbytes` through the communicator's transport, compared word for word."""
        self._check(self.lib.dlv_comm_selftest(self.handle, int(nbytes)))

    def bcast_weights(self, root: int = 0):
        self._check(self.lib.dlv_bcast_weights(self.handle, int(root)))
        for e in self.engines:
            e.features = self.engines[root].features

    def range_recover(self) -> int:
        """dlv_comm_range_recover after sw_infer_sharded raised DLV_ERANGE: the same next block shifts on every rank
        -> number of blocks changed (raises DLV_ERANGE when nothing is left)"""
        n = C.c_int()
        self._check(self.lib.dlv_comm_range_recover(self.handle, C.byref(n)))
        return n.value

    def make_plan(self, params, weights=None):
        from .parallel import plan_from_params

        return plan_from_params(params, len(self.devices), weights)

    def sw_infer_sharded(self, params, plan, slabs, vols, accs, cnts=None):
        """slabs[r] = (z0, nz); vols / accs / cnts: per-rank tensors on devices[r] holding those planes."""
        n = len(self.devices)
        pc = _lib.ShardPlanC()
        pc.world, pc.n_windows = n, plan.n_windows
        for r in range(n):
            pc.win_begin[r], pc.win_end[r] = plan.win_ranges[r]
            pc.z_comp_lo[r], pc.z_comp_hi[r] = plan.z_computed[r]
            pc.z_own_lo[r], pc.z_own_hi[r] = plan.z_owned[r]
        z0 = (C.c_int * n)(*[int(s[0]) for s in slabs])
        nz = (C.c_int * n)(*[int(s[1]) for s in slabs])
        torch = self.torch
        vp = (C.c_void_p * n)(*[self.engines[r]._dev(vols[r], torch.uint16, "vol").value for r in range(n)])
        ap = (C.c_void_p * n)(*[self.engines[r]._dev(accs[r], torch.float32, "acc").value for r in range(n)])
        cp = None
        if cnts is not None:
            cp = (C.c_void_p * n)(*[self.engines[r]._dev(cnts[r], torch.uint8, "cnt").value for r in range(n)])
        st = (_lib.SwStats * n)()
        for d in set(self.devices):
            torch.cuda.synchronize(d)
        self._check(self.lib.dlv_sw_infer_sharded(self.handle, C.byref(params), C.byref(pc), z0, nz, vp, ap, cp, st))
        return [{"n_windows": s.n_windows, "n_skipped": s.n_skipped, "n_forward_launches": s.n_forward_launches} for s in st]


_shared_engines: Dict[int, "HipEngine"] = {}


def shared_engine(device: int = 0) -> "HipEngine":
    """The process-wide engine of a device.  run_inference and count_blobs use it by default, so that one `python -m
    delivr_cfos_amd` run keeps its context between steps and brains: the workspaces of a pass (~35 GB) and of the labelling
    (~2 GB) and the pinned staging ring are allocated once and never handed back to the driver in between (large allocations
    that follow a release took up to seconds on this platform: profiles/r06r_alloc_probe2.json).  The reference's counterpart is PyTorch's caching allocator living as long as the process."""
    eng = _shared_engines.get(int(device))
    if eng is None or eng.ctx is None:
        eng = HipEngine(int(device))
        eng.shared = True
        _shared_engines[int(device)] = eng
    return eng


class HipEngine:
    shared = False  # True: owned by shared_engine() - callers that "own" their engine must not close it

    def __init__(self, device: int = 0, _ctx=None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("delivr_cfos_amd needs an MI355X (torch.cuda.is_available() is False); "
                               "there is no CPU fallback for the HIP path")
        self.torch = torch
        self.lib = _lib.load()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self.features: Optional[Tuple[int, ...]] = None
        self._keep = []
        self._owns_ctx = _ctx is None
        if _ctx is not None:
            # a context owned by a HipComm: it runs on its own stream (dlv_stream); torch work is ordered against it by
            # device-wide synchronisation in _enter/_leave
            self.ctx = _ctx
            self.tstream = torch.cuda.ExternalStream(self.lib.dlv_stream(_ctx), device=self.device)
            return
        self.tstream = torch.cuda.Stream(device=self.device)
        ctx = C.c_void_p()
        rc = self.lib.dlv_ctx_create(self.device_index, C.c_void_p(self.tstream.cuda_stream), C.byref(ctx))
        if rc != 0:
            raise DelivrHipError(rc, "dlv_ctx_create failed")
        self.ctx = ctx

    # ---- plumbing ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "ctx", None) and getattr(self, "_owns_ctx", True):
            self.lib.dlv_ctx_destroy(self.ctx)
        self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise DelivrHipError(rc, self.lib.dlv_last_error(self.ctx).decode(errors="replace"))

    def _enter(self):
        """kernels run on the engine stream: order them after whatever torch queued so far"""
        self.tstream.wait_stream(self.torch.cuda.current_stream(self.device))

    def _leave(self):
        self.torch.cuda.current_stream(self.device).wait_stream(self.tstream)

    def sync(self):
        self._check(self.lib.dlv_sync(self.ctx))

    def _dev(self, t, dtype, name):
        torch = self.torch
        if not isinstance(t, torch.Tensor) or t.device != self.device:
            raise TypeError(f"{name}: expected a torch tensor on {self.device}, got {type(t)} on "
                            f"{getattr(t, 'device', None)}")
        if t.dtype != dtype:
            raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise ValueError(f"{name}: tensor must be contiguous")
        return C.c_void_p(t.data_ptr())

    def to_device(self, arr, dtype=None):
        """numpy / torch-cpu -> tensor in HBM (uint16 volumes go through torch.uint16)."""
        torch = self.torch
        if isinstance(arr, torch.Tensor):
            t = arr
        else:
            a = np.ascontiguousarray(arr)
            if not a.flags.writeable:  # (a read-only memmap: torch.from_numpy on it is undefined behaviour; big volumes
                a = np.array(a)        # go through upload_volume, which never wraps the memmap)
            t = torch.from_numpy(a)
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.to(self.device).contiguous()

    def upload_volume(self, arr, z_lo: int = 0, z_hi: Optional[int] = None, chunk_bytes: int = 64 << 20):
        """Host uint16 volume (numpy array or memmap, (..., Z, Y, X)) -> planes [z_lo, z_hi) in HBM through the pinned staging
        ring of hostio.upload: a memmapped .npy is read with parallel preads straight into pinned memory (no whole-volume
        pageable copy; the reference memmaps the file too, inference/inference.py:234), the copy engine moves the previous
        chunk meanwhile, and a rank of a sharded run reads only the planes of its slab."""
        from . import hostio

        a = arr if isinstance(arr, np.ndarray) else np.asarray(arr)
        lead = a.shape[:-3]
        if any(int(v) != 1 for v in lead):
            raise ValueError("one volume, one channel expected")
        if a.dtype != np.uint16:
            raise TypeError(f"upload_volume: uint16 volume expected, got {a.dtype}")
        view = a.reshape(a.shape[-3:])
        Z, Y, X = (int(v) for v in view.shape)
        z_hi = Z if z_hi is None else int(z_hi)
        out = self.torch.empty(tuple(int(v) for v in lead) + (z_hi - z_lo, Y, X), dtype=self.torch.uint16, device=self.device)
        hostio.upload(self, view[z_lo:z_hi], out=out, chunk_bytes=chunk_bytes, what="h2d_volume")
        return out

    # ---- weights -----------------------------------------------------------------------------------
    def load_state_dict(self, checkpoint) -> None:
        """dlv_unet_load from a MONAI BasicUNet checkpoint (inference/inference.py:190-200,222)."""
        torch = self.torch
        sd = strip_state_dict(checkpoint)
        w = _lib.UnetWeights()
        keep = []

        def ptr(key):
            if key not in sd:
                raise KeyError(f"checkpoint has no '{key}' (expected MONAI BasicUNet names)")
            v = sd[key]
            a = v.detach().cpu().float().contiguous().numpy() if isinstance(v, torch.Tensor) else np.ascontiguousarray(v, np.float32)
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p).value, a.shape

        feats = []
        for i, blk in enumerate(CONV_BLOCKS):
            w.conv_w[i], shp = ptr(f"{blk}.conv.weight")
            w.conv_b[i], _ = ptr(f"{blk}.conv.bias")
            w.norm_g[i], _ = ptr(f"{blk}.adn.N.weight")
            w.norm_b[i], _ = ptr(f"{blk}.adn.N.bias")
            if tuple(shp[2:]) != (3, 3, 3):
                raise ValueError(f"{blk}.conv.weight has shape {shp}")
            feats.append(shp)
        for j, blk in enumerate(DECONV_BLOCKS):
            w.deconv_w[j], shp = ptr(f"{blk}.weight")
            w.deconv_b[j], _ = ptr(f"{blk}.bias")
            if tuple(shp[2:]) != (2, 2, 2):
                raise ValueError(f"{blk}.weight has shape {shp}")
        w.final_w, fshape = ptr("final_conv.weight")
        w.final_b, _ = ptr("final_conv.bias")
        features = (feats[0][0], feats[2][0], feats[4][0], feats[6][0], feats[8][0], feats[16][0])
        if feats[0][1] != 1 or fshape[0] != 1 or fshape[1] != features[5]:
            raise ValueError("only in_channels=1 / out_channels=1 networks are supported (inference.py:190-197)")
        for k in range(6):
            w.features[k] = int(features[k])
        self._enter()
        self._check(self.lib.dlv_unet_load(self.ctx, C.byref(w)))
        self._leave()
        self.features = tuple(int(f) for f in features)

    def weight_blob(self):
        """The packed parameters in HBM as a uint8 torch tensor view (for an RCCL broadcast)."""
        n = C.c_size_t()
        p = C.c_void_p()
        self._check(self.lib.dlv_unet_blob_size(self.ctx, C.byref(n)))
        self._check(self.lib.dlv_unet_blob_dev(self.ctx, C.byref(p)))
        return _tensor_view(self.torch, p.value, n.value, self.device)

    def alloc_weight_blob(self, features: Sequence[int]):
        arr = (C.c_int * 6)(*[int(f) for f in features])
        self._check(self.lib.dlv_unet_alloc_blob(self.ctx, arr))
        self.features = tuple(int(f) for f in features)

    # ---- forward -----------------------------------------------------------------------------------
    def unet_forward(self, x, precision: str = "fp32"):
        """(B,1,d,h,w) fp32 tensor in HBM -> logits (B,1,d,h,w) fp32."""
        torch = self.torch
        if x.dim() != 5 or x.shape[1] != 1:
            raise ValueError("x must be (B,1,d,h,w)")
        out = torch.empty_like(x)
        B, _, d, h, w = x.shape
        self._enter()
        self._check(self.lib.dlv_unet_forward_dev(self.ctx, self._dev(x, torch.float32, "x"),
                                                  self._dev(out, torch.float32, "out"), B, d, h, w,
                                                  PRECISIONS[precision]))
        self._leave()
        return out

    def diag_set(self, name: str, value: int = 1) -> None:
        """a kernel-selection switch of this context (include/delivr_hip_diag.h: dlv_diag_set) - tests and A/B runs; the library
        reads no such switch from the environment"""
        self._check(self.lib.dlv_diag_set(self.ctx, name.encode(), int(value)))

    def set_zm_variant(self, variant: int) -> None:
        """A/B and diagnostic builds of the z-march conv (dlv_debug_set_zm_variant); 0 = default."""
        self._check(self.lib.dlv_debug_set_zm_variant(self.ctx, int(variant)))

    def debug_layer_bf16(self, kind: int, index: int, in1, in2=None, precision: str = "bf16"):
        """test hook (dlv_debug_layer_bf16): one conv block / deconv of the bf16 path on fp32 tensors."""
        torch = self.torch
        B, c1, D, H, W = in1.shape
        c2 = 0 if in2 is None else int(in2.shape[1])
        if kind in (0, 2, 3):
            cout = {i: None for i in range(18)}
            blk_out = [self.features[0], self.features[0], self.features[1], self.features[1], self.features[2],
                       self.features[2], self.features[3], self.features[3], self.features[4], self.features[4],
                       self.features[3], self.features[3], self.features[2], self.features[2], self.features[1],
                       self.features[1], self.features[5], self.features[5]]
            out = torch.empty((B, blk_out[index], D, H, W), dtype=torch.float32, device=self.device)
        else:
            dco = [self.features[4] // 2, self.features[3] // 2, self.features[2] // 2, self.features[1]]
            out = torch.empty((B, dco[index], 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=self.device)
        # (one layer runs in ONE format: "bf16" here is bf16 proper)
        self._check(self.lib.dlv_debug_set_format(self.ctx, PREC_F16 if precision in ("fp16", "f16") else PREC_BF16_ALL))
        self._enter()
        self._check(self.lib.dlv_debug_layer_bf16(
            self.ctx, kind, index, self._dev(in1, torch.float32, "in1"), int(c1),
            self._dev(in2, torch.float32, "in2") if in2 is not None else None, c2,
            self._dev(out, torch.float32, "out"), B, D, H, W))
        self._leave()
        return out

    # ---- sliding window ----------------------------------------------------------------------------
    def make_sw_params(self, padded_shape, roi, overlap=0.5, flip_dim=None, skip_threshold=0, precision="fp16",
                       sw_batch=0, win_range=None, slab=None, repeat=1, blend="constant", sigma_scale=0.125,
                       wsum=None) -> _lib.SwParams:
        p = _lib.SwParams()
        p.Zp, p.Yp, p.Xp = (int(v) for v in padded_shape)
        for k in range(3):
            p.roi[k] = int(roi[k])
        p.overlap = float(overlap)
        p.flip_dim = -1 if flip_dim is None else int(flip_dim)
        p.skip_threshold = int(skip_threshold)
        p.precision = PRECISIONS[precision]
        p.sw_batch = int(sw_batch)
        p.win_begin, p.win_end = (0, 0) if win_range is None else (int(win_range[0]), int(win_range[1]))
        p.z0, p.nz = (0, 0) if slab is None else (int(slab[0]), int(slab[1]))
        p.repeat = int(repeat)
        # "constant" is what the reference does (its mode="gaussian" argument is ignored); "gaussian" = MONAI's
        # importance map as an option, with `wsum` (fp32 tensor shaped like acc, optional) collecting the weight sums
        p.blend_mode = {"constant": 0, "gaussian": 1}[blend]
        p.sigma_scale = float(sigma_scale)
        if wsum is not None:
            if blend != "gaussian":
                raise ValueError("wsum is the float count map of the Gaussian blend")
            p.wsum_dev = self._dev(wsum, self.torch.float32, "wsum").value
            p._keep = wsum
        return p

    def num_windows(self, params) -> int:
        n = C.c_int64()
        rc = self.lib.dlv_sw_num_windows(C.byref(params), C.byref(n))
        if rc != 0:
            raise DelivrHipError(rc, "dlv_sw_num_windows: bad geometry")
        return n.value

    def window_starts(self, params) -> np.ndarray:
        n = self.num_windows(params)
        buf = np.zeros((n, 3), dtype=np.int64)
        rc = self.lib.dlv_sw_window_starts(C.byref(params), buf.ctypes.data_as(C.POINTER(C.c_int64)), n)
        if rc != 0:
            raise DelivrHipError(rc, "dlv_sw_window_starts failed")
        return buf

    def window_max(self, params, vol) -> np.ndarray:
        """Per-window maximum (int32, reference window order): windows with max <= skip_threshold are background."""
        n = self.num_windows(params)
        if params.win_end > 0 or params.win_begin > 0:  # a shard of the window list: out[i] = window win_begin + i
            n = max(min(int(params.win_end) if params.win_end > 0 else n, n) - max(int(params.win_begin), 0), 0)
        out = np.zeros(n, dtype=np.int32)
        if n == 0:
            return out
        self._enter()
        self._check(self.lib.dlv_sw_window_max_dev(self.ctx, C.byref(params), self._dev(vol, self.torch.uint16, "vol"),
                                                   out.ctypes.data_as(C.POINTER(C.c_int32)), n))
        self._leave()
        return out

    def reserve(self, params, stack_shape=None) -> None:
        """dlv_reserve_dev: allocate now what a pass with `params` (and the finalize of `stack_shape`) will ask for.  Meant for a
        second thread while the volume is read; the engine must not be used by another thread meanwhile."""
        Z, Y, X = (int(v) for v in stack_shape) if stack_shape is not None else (0, 0, 0)
        self._check(self.lib.dlv_reserve_dev(self.ctx, C.byref(params), Z, Y, X))

    def sw_infer(self, params, vol, acc, cnt=None) -> dict:
        """One sliding-window pass; vol uint16 (nz,Yp,Xp), acc fp32 and cnt uint8 (optional) are
        mutated in place (inference/sliding_window_inferer.py:232-251)."""
        torch = self.torch
        st = _lib.SwStats()
        self._enter()
        self._check(self.lib.dlv_sw_infer_dev(
            self.ctx, C.byref(params), self._dev(vol, torch.uint16, "vol"), self._dev(acc, torch.float32, "acc"),
            self._dev(cnt, torch.uint8, "cnt") if cnt is not None else None, C.byref(st)))
        self._leave()
        return {"n_windows": st.n_windows, "n_skipped": st.n_skipped, "n_forward_launches": st.n_forward_launches}

    # ---- finalize ----------------------------------------------------------------------------------
    def finalize(self, acc, cnt, raw, stack_shape, threshold=0.5, erode_iters=30, zblock=0, want_prob=False, out=None):
        """-> uint8 (Z,Y,X) binaries [, fp32 sigmoid] (inference/inference.py:285-299, :31-95).  `out`: a uint8 (Z,Y,X) tensor to
        write the mask into (allocated ahead of the passes)."""
        torch = self.torch
        Z, Y, X = (int(v) for v in stack_shape)
        Yp, Xp = int(acc.shape[-2]), int(acc.shape[-1])
        if cnt is not None and cnt.dtype == torch.float32:
            # Gaussian blend: the count map holds weight sums; the mean logit is formed here, the kernel sees no count
            acc = acc / cnt.clamp_min(torch.finfo(torch.float32).tiny)
            cnt = None
        if out is None:
            out = torch.empty((Z, Y, X), dtype=torch.uint8, device=self.device)
        elif tuple(out.shape) != (Z, Y, X):
            raise ValueError(f"finalize: out has shape {tuple(out.shape)}, the stack {(Z, Y, X)}")
        prob = torch.empty((Z, Y, X), dtype=torch.float32, device=self.device) if want_prob else None
        self._enter()
        self._check(self.lib.dlv_finalize_dev(
            self.ctx, self._dev(acc, torch.float32, "acc"),
            self._dev(cnt, torch.uint8, "cnt") if cnt is not None else None, self._dev(raw, torch.uint16, "raw"),
            Yp, Xp, Z, Y, X, float(threshold), int(erode_iters), int(zblock),
            self._dev(out, torch.uint8, "out"), self._dev(prob, torch.float32, "prob") if want_prob else None))
        self._leave()
        return (out, prob) if want_prob else out

    def finalize_slab(self, acc, cnt, raw, z_abs0: int, stack_yx, threshold=0.5, erode_iters=30, zblock=0, want_prob=False):
        """finalize on a Z-slab: acc / cnt / raw hold planes [z_abs0, z_abs0 + nz) (nz = acc.shape[0]); the erosion's z-blocks
        sit at absolute multiples of zblock (dlv_finalize_slab_dev).  -> uint8 (nz, Y, X) [, fp32 sigmoid]."""
        torch = self.torch
        nz = int(acc.shape[0])
        Y, X = (int(v) for v in stack_yx)
        Yp, Xp = int(acc.shape[-2]), int(acc.shape[-1])
        if cnt is not None and cnt.dtype == torch.float32:
            acc = acc / cnt.clamp_min(torch.finfo(torch.float32).tiny)
            cnt = None
        out = torch.empty((nz, Y, X), dtype=torch.uint8, device=self.device)
        prob = torch.empty((nz, Y, X), dtype=torch.float32, device=self.device) if want_prob else None
        self._enter()
        self._check(self.lib.dlv_finalize_slab_dev(
            self.ctx, self._dev(acc, torch.float32, "acc"), self._dev(cnt, torch.uint8, "cnt") if cnt is not None else None,
            self._dev(raw, torch.uint16, "raw"), Yp, Xp, int(z_abs0), nz, Y, X, float(threshold), int(erode_iters), int(zblock),
            self._dev(out, torch.uint8, "out"), self._dev(prob, torch.float32, "prob") if want_prob else None))
        self._leave()
        return (out, prob) if want_prob else out

    # ---- connected components ----------------------------------------------------------------------
    def ccl26(self, mask):
        """uint8 (Z,Y,X) in HBM -> (labels uint32 tensor in HBM (stored in an int32 tensor), N)."""
        torch = self.torch
        Z, Y, X = (int(v) for v in mask.shape)
        labels = torch.empty((Z, Y, X), dtype=torch.int32, device=self.device)  # uint32 payload
        n = C.c_uint64()
        self._enter()
        self._check(self.lib.dlv_ccl26_dev(self.ctx, self._dev(mask, torch.uint8, "mask"), Z, Y, X,
                                           C.c_void_p(labels.data_ptr()), C.byref(n)))
        self._leave()
        return labels, int(n.value)

    def cc_stats(self, labels, n: int) -> dict:
        """cc3d.statistics(no_slice_conversion=True) layout (count_blobs.py:85)."""
        Z, Y, X = (int(v) for v in labels.shape)
        counts = np.zeros(n + 1, dtype=np.uint32)
        bbox = np.zeros((n + 1, 6), dtype=np.uint16)
        cent = np.zeros((n + 1, 3), dtype=np.float64)
        self._enter()
        self._check(self.lib.dlv_cc_stats_dev(self.ctx, C.c_void_p(labels.data_ptr()), Z, Y, X, n,
                                              counts.ctypes.data_as(C.c_void_p), bbox.ctypes.data_as(C.c_void_p),
                                              cent.ctypes.data_as(C.c_void_p)))
        self._leave()
        return {"voxel_counts": counts, "bounding_boxes": bbox, "centroids": cent}

    def cc_stats_raw(self, labels, n: int) -> dict:
        """Accumulators behind cc_stats (dlv_cc_stats_raw_dev): counts u32, bbmin/bbmax u32 (n+1,3), sums u64 (n+1,3)."""
        Z, Y, X = (int(v) for v in labels.shape)
        counts = np.zeros(n + 1, dtype=np.uint32)
        bbmin = np.zeros((n + 1, 3), dtype=np.uint32)
        bbmax = np.zeros((n + 1, 3), dtype=np.uint32)
        sums = np.zeros((n + 1, 3), dtype=np.uint64)
        self._enter()
        self._check(self.lib.dlv_cc_stats_raw_dev(self.ctx, C.c_void_p(labels.data_ptr()), Z, Y, X, n,
                                                  counts.ctypes.data_as(C.c_void_p), bbmin.ctypes.data_as(C.c_void_p),
                                                  bbmax.ctypes.data_as(C.c_void_p), sums.ctypes.data_as(C.c_void_p)))
        self._leave()
        return {"counts": counts, "bbmin": bbmin, "bbmax": bbmax, "sums": sums}

    def seam_pairs(self, plane_a, plane_b) -> np.ndarray:
        """Unique (label in plane_a, label in plane_b) pairs that are 26-adjacent across a slab seam; (k,2) uint32,
        sorted (dlv_seam_pairs_dev: count pass, then emit pass)."""
        torch = self.torch
        Y, X = (int(v) for v in plane_a.shape)
        a = self._dev(plane_a, torch.int32, "plane_a")
        b = self._dev(plane_b, torch.int32, "plane_b")
        cnt = C.c_uint64()
        self._enter()
        self._check(self.lib.dlv_seam_pairs_dev(self.ctx, a, b, Y, X, None, 0, C.byref(cnt)))
        k = int(cnt.value)
        if k == 0:
            self._leave()
            return np.zeros((0, 2), dtype=np.uint32)
        pairs = torch.empty((k, 2), dtype=torch.int32, device=self.device)
        self._check(self.lib.dlv_seam_pairs_dev(self.ctx, a, b, Y, X, C.c_void_p(pairs.data_ptr()), k, C.byref(cnt)))
        self._leave()
        assert int(cnt.value) == k
        return np.unique(pairs.cpu().numpy().view(np.uint32), axis=0)

    def relabel(self, labels, lut: np.ndarray) -> None:
        """labels[i] = lut[labels[i]] in place (dlv_relabel_u32_dev); lut: uint32 (n_local+1,), lut[0] = 0."""
        torch = self.torch
        lut_dev = torch.from_numpy(np.ascontiguousarray(lut, dtype=np.uint32).view(np.int32)).to(self.device)
        self._enter()
        self._check(self.lib.dlv_relabel_u32_dev(self.ctx, C.c_void_p(labels.data_ptr()), int(labels.numel()),
                                                 C.c_void_p(lut_dev.data_ptr()), int(lut_dev.numel())))
        self._leave()
        self.sync()

    # ---- blob painting -----------------------------------------------------------------------------
    def edt_u16(self, stack, sampling_zyx):
        """blob_depthmap.py:160-170: exact Euclidean distance (units of `sampling_zyx`) of every non-zero voxel of the
        uint16 stack to the nearest zero voxel, the stack surrounded by zeros; truncated to uint16 (dlv_edt_u16_dev)."""
        torch = self.torch
        Z, Y, X = (int(v) for v in stack.shape)
        out = torch.empty((Z, Y, X), dtype=torch.uint16, device=self.device)
        samp = (C.c_double * 3)(*[float(v) for v in sampling_zyx])
        self._enter()
        self._check(self.lib.dlv_edt_u16_dev(self.ctx, self._dev(stack, torch.uint16, "stack"), Z, Y, X, samp,
                                             C.c_void_p(out.data_ptr())))
        self._leave()
        return out

    def paint_boxes(self, bin_img, boxes: np.ndarray, values):
        """blob_highlighter.py:108-125 / :150-158 on the device.  bin_img: uint8 (Z,Y,X) in HBM; boxes (n,6) int32
        half-open slices in painting order; values: list of (n,) uint8 / uint16 arrays (one image per array).
        Returns the list of painted images (HBM tensors)."""
        torch = self.torch
        Z, Y, X = (int(v) for v in bin_img.shape)
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 6)
        n = int(boxes.shape[0])
        owner = torch.empty((Z, Y, X), dtype=torch.int32, device=self.device)
        boxes_dev = torch.from_numpy(boxes).to(self.device) if n else None
        self._enter()
        self._check(self.lib.dlv_paint_owner_dev(
            self.ctx, self._dev(bin_img, torch.uint8, "bin_img"), Z, Y, X,
            C.c_void_p(boxes_dev.data_ptr()) if n else None, boxes.ctypes.data_as(C.c_void_p) if n else None, n,
            C.c_void_p(owner.data_ptr())))
        outs = []
        for val in values:
            val = np.ascontiguousarray(val)
            if val.dtype not in (np.uint8, np.uint16) or val.shape != (n,):
                raise TypeError("values: (n_boxes,) uint8 or uint16 arrays")
            tdt = torch.uint8 if val.dtype == np.uint8 else torch.uint16
            vdev = torch.from_numpy(val if n else np.zeros(1, val.dtype)).to(self.device)
            out = torch.empty((Z, Y, X), dtype=tdt, device=self.device)
            self._check(self.lib.dlv_paint_apply_dev(self.ctx, C.c_void_p(owner.data_ptr()), C.c_void_p(bin_img.data_ptr()),
                                                     Z * Y * X, C.c_void_p(vdev.data_ptr()), val.dtype.itemsize,
                                                     C.c_void_p(out.data_ptr())))
            outs.append(out)
        self._leave()
        self.sync()
        return outs

    # ---- atlas-space heat map ----------------------------------------------------------------------
    def heatmap(self, cells_xyz: np.ndarray, shape_zyx, sigma: float = 2.25):
        """create_heatmap (cells_to_atlas.py:174-200) on the device: float32 (Z,Y,X) tensor in HBM."""
        torch = self.torch
        Z, Y, X = (int(v) for v in shape_zyx)
        xyz = np.ascontiguousarray(cells_xyz, dtype=np.int32).reshape(-1, 3)
        n = int(xyz.shape[0])
        heat = torch.empty((Z, Y, X), dtype=torch.float32, device=self.device)
        tmp = torch.empty_like(heat)
        xyz_dev = torch.from_numpy(xyz).to(self.device) if n else None
        # scipy.ndimage._filters._gaussian_kernel1d, order 0
        radius = int(4.0 * float(sigma) + 0.5)
        x = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
        phi = phi / phi.sum()
        w = np.ascontiguousarray(phi[radius:], dtype=np.float64)
        self._enter()
        self._check(self.lib.dlv_heatmap_counts_dev(self.ctx, C.c_void_p(xyz_dev.data_ptr()) if n else None, n, Z, Y, X,
                                                    C.c_void_p(heat.data_ptr())))
        self._check(self.lib.dlv_gauss_blur_f32_dev(self.ctx, C.c_void_p(heat.data_ptr()), Z, Y, X,
                                                    w.ctypes.data_as(C.c_void_p), radius, C.c_void_p(tmp.data_ptr())))
        self._leave()
        return heat

    # ---- resamplers --------------------------------------------------------------------------------
    def block_mean_u16(self, vol, factors):
        torch = self.torch
        Z, Y, X = (int(v) for v in vol.shape)
        fz, fy, fx = (int(f) for f in factors)
        out = torch.empty((-(-Z // fz), -(-Y // fy), -(-X // fx)), dtype=torch.uint16, device=self.device)
        self._enter()
        self._check(self.lib.dlv_block_mean_u16_dev(self.ctx, self._dev(vol, torch.uint16, "vol"), Z, Y, X, fz, fy, fx,
                                                    self._dev(out, torch.uint16, "out")))
        self._leave()
        return out

    def zoom_spline2_u8(self, mask, out_shape):
        torch = self.torch
        iz, iy, ix = (int(v) for v in mask.shape)
        oz, oy, ox = (int(v) for v in out_shape)
        out = torch.empty((oz, oy, ox), dtype=torch.uint8, device=self.device)
        self._enter()
        self._check(self.lib.dlv_zoom_spline2_u8_dev(self.ctx, self._dev(mask, torch.uint8, "mask"), iz, iy, ix,
                                                     self._dev(out, torch.uint8, "out"), oz, oy, ox))
        self._leave()
        return out

    def mask_pad_u16(self, raw, mask, padded_shape, threshold=0):
        torch = self.torch
        Z, Y, X = (int(v) for v in raw.shape)
        Zp, Yp, Xp = (int(v) for v in padded_shape)
        out = torch.empty((Zp, Yp, Xp), dtype=torch.uint16, device=self.device)
        self._enter()
        self._check(self.lib.dlv_mask_pad_u16_dev(
            self.ctx, self._dev(raw, torch.uint16, "raw"),
            self._dev(mask, torch.uint8, "mask") if mask is not None else None, int(threshold), Z, Y, X,
            self._dev(out, torch.uint16, "out"), Zp, Yp, Xp))
        self._leave()
        return out

    def trilinear_u16(self, vol, out_shape):
        torch = self.torch
        iz, iy, ix = (int(v) for v in vol.shape)
        oz, oy, ox = (int(v) for v in out_shape)
        out = torch.empty((oz, oy, ox), dtype=torch.uint16, device=self.device)
        self._enter()
        self._check(self.lib.dlv_trilinear_u16_dev(self.ctx, self._dev(vol, torch.uint16, "vol"), iz, iy, ix,
                                                   self._dev(out, torch.uint16, "out"), oz, oy, ox))
        self._leave()
        return out

    def affine_warp_u16(self, vol, matrix34, out_shape):
        """uint16 (Z,Y,X) in HBM resampled through a 3x4 affine map (output voxel index -> input voxel index, index
        space, zero outside, trilinear, round half up) - the north-star's atlas-space warp (dlv_affine_warp_u16_dev)."""
        torch = self.torch
        iz, iy, ix = (int(v) for v in vol.shape)
        oz, oy, ox = (int(v) for v in out_shape)
        m = np.ascontiguousarray(np.asarray(matrix34, dtype=np.float64).reshape(12))
        out = torch.empty((oz, oy, ox), dtype=torch.uint16, device=self.device)
        self._enter()
        self._check(self.lib.dlv_affine_warp_u16_dev(self.ctx, self._dev(vol, torch.uint16, "vol"), iz, iy, ix,
                                                     m.ctypes.data_as(C.POINTER(C.c_double)),
                                                     self._dev(out, torch.uint16, "out"), oz, oy, ox))
        self._leave()
        return out

    # ---- kernel timer ------------------------------------------------------------------------------
    # ---- fp16 range guard: per-block output shift (dlv_unet_set_conv_shift) ----------------------------------------
    def set_conv_shift(self, layer: int, shift: int) -> None:
        """conv block `layer` (0..17) stores its raw output 2^-shift times smaller (16-bit paths only; InstanceNorm removes the
        factor exactly)"""
        self._enter()
        self._check(self.lib.dlv_unet_set_conv_shift(self.ctx, int(layer), int(shift)))
        self._leave()

    def conv_shifts(self):
        out = []
        for i in range(_lib.N_CONV):
            v = C.c_int()
            self._check(self.lib.dlv_unet_get_conv_shift(self.ctx, i, C.byref(v)))
            out.append(v.value)
        return out

    def note_conv_shifts(self, shifts) -> None:
        """the blob this engine received by broadcast was packed with these per-block shifts (no repack)"""
        arr = (C.c_int * _lib.N_CONV)(*[int(v) for v in shifts])
        self._check(self.lib.dlv_unet_note_conv_shifts(self.ctx, arr))

    def range_report(self):
        """(layer the last DLV_ERANGE named or -1, [|mean| + 8 sigma of every conv block's raw output where it exceeded 4096])"""
        layer = C.c_int()
        peaks = (C.c_float * _lib.N_CONV)()
        self._check(self.lib.dlv_range_report(self.ctx, C.byref(layer), peaks))
        return layer.value, [float(v) for v in peaks]

    def range_recover(self) -> int:
        """dlv_range_recover: the library's own next step after DLV_ERANGE (what range_guard.next_shifts decides, applied);
        -> number of conv blocks whose shift changed; raises DelivrHipError(DLV_ERANGE) when nothing is left to try"""
        n = C.c_int()
        self._enter()
        try:
            self._check(self.lib.dlv_range_recover(self.ctx, C.byref(n)))
        finally:
            self._leave()
        return n.value

    def set_lanes(self, lanes: int):
        self._check(self.lib.dlv_set_lanes(self.ctx, int(lanes)))

    def prof_enable(self, on: bool = True):
        self._check(self.lib.dlv_prof_enable(self.ctx, 1 if on else 0))

    def prof_reset(self):
        self._check(self.lib.dlv_prof_reset(self.ctx))

    def prof_report(self) -> Dict[str, dict]:
        ent = (_lib.ProfEntry * _lib.PROF_MAX)()
        n = C.c_int()
        self._check(self.lib.dlv_prof_report(self.ctx, ent, _lib.PROF_MAX, C.byref(n)))
        out = {}
        for i in range(min(n.value, _lib.PROF_MAX)):
            e = ent[i]
            out[e.name.decode()] = {"launches": e.launches, "total_ms": e.total_ms, "flops": e.flops, "bytes": e.bytes}
        return out


def _tensor_view(torch, ptr: int, nbytes: int, device):
    """uint8 torch view of library-owned HBM (no copy) via __cuda_array_interface__."""

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)
