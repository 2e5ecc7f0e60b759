"""The hot path as the face model runs it (third piece): FaceTargetNet.forward evaluates TWO
ExtractorAttn per attention layer on the same decoder features -- `attn_p` warps the previous frame's features, `attn_r`
the reference frame's -- and blends them by their masks (reference: model/networks/generator.py:490-499); FaceGenerator
generates the frames of a clip one after the other because frame t's image is frame t+1's "previous" input (:402-426).

What that recurrence leaves to overlap is exactly one thing: inside a frame the two blocks of a layer are independent of
each other (both read `out`, neither reads the other's result until `out_p + out_r`).  At the face model's batch (clips of
6 frames, a handful of clips per GPU) an ExtractorAttn call is ~100 short launches, so the two chains are issued on two
HIP streams and interleave on the chip:

    cur:  ... out ─┬─ attn_p(prev, out, flow_p) ──────────┬─ blend (one kernel) ─ decoder ...
                   └─ (side) attn_r(ref, out, flow_r) ────┘

`DualStreamAttn` is that fork / join with the event dependencies spelled out (forward; autograd replays each node on
the stream it ran on and inserts the reverse dependencies itself).  The blend behind the pair -- nine launch-sized
elementwise kernels forward, ~14 backward in the op-by-op evaluation -- is one kernel each way (`MaskBlendFunction`,
csrc/mask_blend.hip), bit-identical in the forward.  `face_target_forward` is FaceTargetNet.forward with
the pair routed through it; `install(..., dual_stream_face=True)` patches it into the reference's class, leaving
__init__, attribute names and state_dict keys alone.  Frames stay sequential, as in the reference.
"""
import torch
from torch.autograd import Function

from . import _lib


class MaskBlendFunction(Function):
    """(out*(1-mask_p) + attn_p*mask_p) + (out*(1-mask_r) + attn_r*mask_r)  -- generator.py:496-499 -- as one kernel each way
    (csrc/mask_blend.hip) instead of nine elementwise kernels forward and ~14 backward.  Forward: the op-by-op result bit for
    bit (intermediates rounded to the storage type where torch rounds them); mask gradients are sums over the channels,
    accumulated in float32."""

    @staticmethod
    def forward(ctx, out, attn_p, attn_r, mask_p, mask_r):
        # the kernel takes raw pointers and B, C, H*W from `out` alone: anything that is not exactly that layout would be
        # read out of bounds or reinterpreted, so it is refused here (DualStreamAttn falls back to the op-by-op blend)
        if out.dim() != 4:
            raise ValueError("MaskBlendFunction: out must be (B, C, H, W), got %s" % (tuple(out.shape),))
        for name, t in (("attn_p", attn_p), ("attn_r", attn_r)):
            if t.shape != out.shape:
                raise ValueError("MaskBlendFunction: %s %s must have the shape of out %s" % (name, tuple(t.shape), tuple(out.shape)))
        want = (out.size(0), 1, out.size(2), out.size(3))
        for name, t in (("mask_p", mask_p), ("mask_r", mask_r)):
            if tuple(t.shape) != want:
                raise ValueError("MaskBlendFunction: %s %s must be %s (no broadcasting here)" % (name, tuple(t.shape), want))
        for name, t in (("attn_p", attn_p), ("attn_r", attn_r), ("mask_p", mask_p), ("mask_r", mask_r)):
            if t.dtype != out.dtype:
                raise TypeError("MaskBlendFunction: %s is %s, out is %s" % (name, t.dtype, out.dtype))
        _lib.require_gpu(out, attn_p, attn_r, mask_p, mask_r)
        out, attn_p, attn_r = out.contiguous(), attn_p.contiguous(), attn_r.contiguous()
        mask_p, mask_r = mask_p.contiguous(), mask_r.contiguous()
        B, C, H, W = out.shape
        sfx = _lib.suffix(out, "mask_blend")
        y = torch.empty_like(out)
        _lib.call("gfla_mask_blend_fwd_" + sfx, out, _lib.ptr(out), _lib.ptr(attn_p), _lib.ptr(attn_r), _lib.ptr(mask_p),
                  _lib.ptr(mask_r), _lib.ptr(y), B, C, H * W)
        ctx.save_for_backward(out, attn_p, attn_r, mask_p, mask_r)
        return y

    @staticmethod
    def backward(ctx, g):
        out, attn_p, attn_r, mask_p, mask_r = ctx.saved_tensors
        B, C, H, W = out.shape
        need = ctx.needs_input_grad
        g = g.to(out.dtype).contiguous()
        new = lambda t, wanted: torch.empty_like(t) if wanted else None
        g_out, g_ap, g_ar = new(out, need[0]), new(attn_p, need[1]), new(attn_r, need[2])
        zeros32 = lambda t, wanted: torch.zeros(t.shape, dtype=torch.float32, device=t.device) if wanted else None
        g_mp, g_mr = zeros32(mask_p, need[3]), zeros32(mask_r, need[4])
        _lib.call("gfla_mask_blend_bwd_" + _lib.suffix(out, "mask_blend"), out, _lib.ptr(out), _lib.ptr(attn_p),
                  _lib.ptr(attn_r), _lib.ptr(mask_p), _lib.ptr(mask_r), _lib.ptr(g), _lib.ptr(g_out), _lib.ptr(g_ap),
                  _lib.ptr(g_ar), _lib.ptr(g_mp), _lib.ptr(g_mr), B, C, H * W)
        cast = lambda t, like: None if t is None else t.to(like.dtype)
        return g_out, g_ap, g_ar, cast(g_mp, mask_p), cast(g_mr, mask_r)


def _blend_fusable(out, attn_p, attn_r, mask_p, mask_r):
    return (out.is_cuda and out.dtype in (torch.float32, torch.bfloat16) and out.dim() == 4
            and attn_p.shape == out.shape == attn_r.shape and attn_p.dtype == out.dtype == attn_r.dtype
            and mask_p.dtype == out.dtype == mask_r.dtype
            and tuple(mask_p.shape) == (out.size(0), 1, out.size(2), out.size(3)) == tuple(mask_r.shape))


_SIDE_STREAMS = {}   # one side stream per device for the whole process (replicas of a DataParallel module each find their own)


def side_stream(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class DualStreamAttn(object):
    """out*(1-m_p) + attn_p(prev, out, flow_p)*m_p  +  out*(1-m_r) + attn_r(ref, out, flow_r)*m_r   (generator.py:494-499)
    with the reference-frame half on a side stream.  enabled=False (or CPU tensors) evaluates the same expression
    sequentially on the current stream -- the parity tests compare the two.  Holds the two modules it was given and
    nothing else; face_target_forward builds one per call from the module's CURRENT attributes."""

    def __init__(self, attn_p, attn_r, enabled=True, fused_blend=True):
        self.attn_p, self.attn_r, self.enabled, self.fused_blend = attn_p, attn_r, enabled, fused_blend

    @staticmethod
    def side_stream(device):
        return side_stream(device)

    @staticmethod
    def _blend(out, attn, mask):
        return out * (1 - mask) + attn * mask

    def __call__(self, out, prev_feature, ref_feature, flow_p, flow_r, mask_p, mask_r):
        if not (self.enabled and out.is_cuda):
            a_p, a_r = self.attn_p(prev_feature, out, flow_p), self.attn_r(ref_feature, out, flow_r)
            if self.fused_blend and _blend_fusable(out, a_p, a_r, mask_p, mask_r):
                return MaskBlendFunction.apply(out, a_p, a_r, mask_p, mask_r)
            return self._blend(out, a_p, mask_p) + self._blend(out, a_r, mask_r)
        cur = torch.cuda.current_stream(out.device)
        side = self.side_stream(out.device)
        ready = cur.record_event()                 # everything the side chain reads has been enqueued on `cur`
        with torch.cuda.stream(side):
            side.wait_event(ready)
            a_r = self.attn_r(ref_feature, out, flow_r)
            done = side.record_event()
        # tensors that cross streams: tell the caching allocator who else uses them
        for t in (out, ref_feature, flow_r):
            t.record_stream(side)
        a_p = self.attn_p(prev_feature, out, flow_p)
        cur.wait_event(done)
        a_r.record_stream(cur)
        if self.fused_blend and _blend_fusable(out, a_p, a_r, mask_p, mask_r):
            return MaskBlendFunction.apply(out, a_p, a_r, mask_p, mask_r)
        return self._blend(out, a_p, mask_p) + self._blend(out, a_r, mask_r)


def face_target_forward(self, BP, previous_feature_list, reference_feature_list, flow_fields, masks):
    """FaceTargetNet.forward (generator.py:480-505) with each layer's (attn_p, attn_r) pair on two streams.  `self` is the
    reference's module: block0 / encoder<i> / decoder<i> / attn_p<i> / attn_r<i> / outconv are its attributes."""
    out = self.block0(BP)
    for i in range(self.layers - 1):
        out = getattr(self, "encoder" + str(i))(out)
    counter = 0
    for i in range(self.layers):
        if self.layers - i in self.attn_layer:
            # resolved on every call: a replica made by nn.DataParallel.replicate (a shallow copy of __dict__) must use ITS
            # attn modules and device, a module swapped in later must be seen, and `dual_stream` can be toggled any time
            pair = DualStreamAttn(getattr(self, "attn_p" + str(i)), getattr(self, "attn_r" + str(i)),
                                  getattr(self, "dual_stream", True))
            out = pair(out, previous_feature_list[i], reference_feature_list[i], flow_fields[2 * counter],
                       flow_fields[2 * counter + 1], masks[2 * counter], masks[2 * counter + 1])
            counter += 1
        out = getattr(self, "decoder" + str(i))(out)
    return self.outconv(out)


def patch_reference_face_target_net(cls):
    """Swap `face_target_forward` into an externally defined FaceTargetNet class (the reference's, imported unchanged)."""
    cls.forward = face_target_forward
    return cls


def generate_frames(frame_fn, n_frames, previous, reference):
    """The recurrence of FaceGenerator.forward (generator.py:406-426) around any per-frame callable:
    `frame_fn(t, previous, reference) -> image` ; frame t's image is frame t+1's `previous` (the first frame's previous is
    the reference when none is given).  Frames are issued in order on the current stream -- there is no independent work
    between them to overlap."""
    images = []
    previous = reference if previous is None else previous
    for t in range(n_frames):
        image = frame_fn(t, previous, reference)
        images.append(image)
        previous = image
    return images
