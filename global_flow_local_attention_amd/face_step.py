"""The hot path as the face model runs it (third piece): FaceTargetNet.forward evaluates TWO
ExtractorAttn per attention layer on the same decoder features -- `attn_p` warps the previous frame's features, `attn_r`
the reference frame's -- and blends them by their masks (reference: model/networks/generator.py:490-499); FaceGenerator
generates the frames of a clip one after the other because frame t's image is frame t+1's "previous" input (:402-426).

What that recurrence leaves to overlap is exactly one thing: inside a frame the two blocks of a layer are independent of
each other (both read `out`, neither reads the other's result until `out_p + out_r`).  At the face model's batch (clips of
6 frames, a handful of clips per GPU) an ExtractorAttn call is ~100 short launches, so the two chains are issued on two
HIP streams and interleave on the chip:

    cur:  ... out ─┬─ attn_p(prev, out, flow_p) ─ blend_p ─┬─ out_p + out_r ─ decoder ...
                   └─ (side) attn_r(ref, out, flow_r) ─ blend_r ─┘

`DualStreamAttn` is that fork / join with the event dependencies spelled out (forward; autograd replays each node on
the stream it ran on and inserts the reverse dependencies itself).  `face_target_forward` is FaceTargetNet.forward with
the pair routed through it; `install(..., dual_stream_face=True)` patches it into the reference's class, leaving
__init__, attribute names and state_dict keys alone.  Frames stay sequential, as in the reference.
"""
import torch


class DualStreamAttn(object):
    """out*(1-m_p) + attn_p(prev, out, flow_p)*m_p  +  out*(1-m_r) + attn_r(ref, out, flow_r)*m_r   (generator.py:494-499)
    with the reference-frame half on a side stream.  enabled=False (or CPU tensors) evaluates the same expression
    sequentially on the current stream -- the parity tests compare the two."""

    def __init__(self, attn_p, attn_r, enabled=True):
        self.attn_p, self.attn_r, self.enabled = attn_p, attn_r, enabled
        self._side = None

    def side_stream(self, device):
        if self._side is None or self._side.device != device:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    @staticmethod
    def _blend(out, attn, mask):
        return out * (1 - mask) + attn * mask

    def __call__(self, out, prev_feature, ref_feature, flow_p, flow_r, mask_p, mask_r):
        if not (self.enabled and out.is_cuda):
            out_p = self._blend(out, self.attn_p(prev_feature, out, flow_p), mask_p)
            out_r = self._blend(out, self.attn_r(ref_feature, out, flow_r), mask_r)
            return out_p + out_r
        cur = torch.cuda.current_stream(out.device)
        side = self.side_stream(out.device)
        ready = cur.record_event()                 # everything the side chain reads has been enqueued on `cur`
        with torch.cuda.stream(side):
            side.wait_event(ready)
            out_r = self._blend(out, self.attn_r(ref_feature, out, flow_r), mask_r)
            done = side.record_event()
        # tensors that cross streams: tell the caching allocator who else uses them
        for t in (out, ref_feature, flow_r, mask_r):
            t.record_stream(side)
        out_p = self._blend(out, self.attn_p(prev_feature, out, flow_p), mask_p)
        cur.wait_event(done)
        out_r.record_stream(cur)
        return out_p + out_r


def face_target_forward(self, BP, previous_feature_list, reference_feature_list, flow_fields, masks):
    """FaceTargetNet.forward (generator.py:480-505) with each layer's (attn_p, attn_r) pair on two streams.  `self` is the
    reference's module: block0 / encoder<i> / decoder<i> / attn_p<i> / attn_r<i> / outconv are its attributes."""
    out = self.block0(BP)
    for i in range(self.layers - 1):
        out = getattr(self, "encoder" + str(i))(out)
    pairs = self.__dict__.setdefault("_gfla_pairs", {})
    counter = 0
    for i in range(self.layers):
        if self.layers - i in self.attn_layer:
            pair = pairs.get(i)
            if pair is None:
                pair = pairs[i] = DualStreamAttn(getattr(self, "attn_p" + str(i)), getattr(self, "attn_r" + str(i)),
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
