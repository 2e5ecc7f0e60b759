"""One-process-per-GPU training shell around the UNMODIFIED reference generators.

The reference trains through `BaseModel` (model/base_model.py) + `nn.DataParallel` (model/face_model.py:81-93):
one Python process scatters the batch over the GPUs, gathers outputs and reduces gradients through GPU 0.  Here every
GPU has its own process (torch.distributed over RCCL), owns a slice of the batch, and the gradients are averaged by a
bucketed all-reduce launched from autograd hooks (dist.GradBucketReducer), overlapping the rest of backward.

* `load_reference_checkpoint(net, path)` -- the outcome of `BaseModel.load_networks` (base_model.py:154-197) as a key
  normalisation: strip / add the `module.` prefix DataParallel checkpoints carry, intersect with the network's keys,
  keep shape-matching entries; returns the top-level submodules that kept values of their own.
* `TrainerShell` -- one `optimize_parameters` step (pose_model.py:186-196) for one rank: shard, forward, loss terms,
  backward (reducer hooks), optimizer step.  The adversarial and VGG style/content terms need the reference's
  discriminators / a pretrained VGG19 and are injected as callables (stubbed to zero by default, as BASELINE config 4
  prescribes); the L1 reconstruction, sampling-correctness and affine-regularisation terms use this package's ops.
"""
import torch
import torch.nn as nn

from . import dist as gdist


_DP_PREFIX = "module."   # what nn.DataParallel / DistributedDataParallel put in front of every key


def _strip_prefix(key):
    return key[len(_DP_PREFIX):] if key.startswith(_DP_PREFIX) else key


def load_reference_checkpoint(net, path_or_state, map_location="cpu"):
    """Load a reference `<epoch>_net_G.pth` into `net` with the outcome `BaseModel.load_networks` has
    (base_model.py:154-197): an exact checkpoint loads as is; a checkpoint written from (or into) a DataParallel wrapper
    is matched modulo the `module.` prefix; surplus checkpoint entries are ignored; entries whose shape differs from the
    network's, and network entries the checkpoint lacks, keep the network's own values.  Returns the sorted top-level
    submodule names that kept at least one own value (empty list = every tensor came from the checkpoint).

    Implemented as key normalisation: both key sets are reduced to their prefix-free form, intersected, filtered by
    shape, and the survivors are written into the network's own state dict."""
    state = torch.load(path_or_state, map_location=map_location) if isinstance(path_or_state, str) else path_or_state
    own = net.state_dict()
    # network key -> checkpoint tensor, matching on the prefix-free name; an exact-name match wins over a normalised one
    # (the reference only falls back to prefix juggling when no key matches literally)
    by_plain = {}
    for key, tensor in state.items():
        by_plain.setdefault(_strip_prefix(key), tensor)
    literal_hits = sum(1 for key in own if key in state)
    chosen, kept_own = {}, set()
    for key, current in own.items():
        if literal_hits:
            candidate = state.get(key)
        else:
            candidate = by_plain.get(_strip_prefix(key))
        if candidate is not None and tuple(candidate.shape) == tuple(current.shape):
            chosen[key] = candidate
        else:
            kept_own.add(_strip_prefix(key).split(".")[0])
    merged = dict(own)
    merged.update(chosen)
    net.load_state_dict(merged)
    return sorted(kept_own)


class TrainerShell(object):
    """net_G: the reference generator (e.g. PoseGenerator built after gfla.install()); its forward is called as
    net_G(*inputs) and must return (generated, flow_fields, masks) like generator.py:13-36.
    lambdas: weights of the loss terms (pose_model.py:34-41 defaults).  gan_loss / style_content_loss: callables
    (generated, target) -> scalar; None = 0 (stubbed)."""

    def __init__(self, net_G, lr=1e-4, betas=(0.0, 0.999), lambda_rec=5.0, lambda_correct=5.0, lambda_regularization=0.0025,
                 correctness=None, regularization=None, gan_loss=None, style_content_loss=None, attn_layer=(2, 3),
                 bucket_mb=32.0):
        self.net_G = net_G
        self.optimizer_G = torch.optim.Adam([p for p in net_G.parameters() if p.requires_grad], lr=lr, betas=betas)
        self.l1 = nn.L1Loss()
        self.lambdas = dict(rec=lambda_rec, correct=lambda_correct, regularization=lambda_regularization)
        self.correctness, self.regularization = correctness, regularization
        self.gan_loss, self.style_content_loss = gan_loss, style_content_loss
        self.attn_layer = list(attn_layer)
        self.reducer = gdist.GradBucketReducer(list(net_G.parameters()), bucket_mb=bucket_mb)
        self.losses = {}

    def shard(self, *tensors):
        """This rank's slice of a global batch (DataParallel's scatter along dim 0)."""
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return tensors
        r, w = torch.distributed.get_rank(), torch.distributed.get_world_size()
        return tuple(gdist.shard_batch(t, r, w) for t in tensors)

    def optimize_parameters(self, inputs, target, source=None):
        """One generator step on THIS rank's (already sharded) batch.  Returns the loss terms as floats."""
        self.optimizer_G.zero_grad(set_to_none=True)
        generated, flow_fields, _masks = self.net_G(*inputs)
        terms = {"app_gen": self.l1(generated, target) * self.lambdas["rec"]}                       # pose_model.py:165
        if self.correctness is not None and source is not None:                                      # :158-159
            terms["correctness_gen"] = self.correctness(target, source, flow_fields, self.attn_layer) * self.lambdas["correct"]
        if self.regularization is not None:                                                          # :161-162
            terms["regularization"] = self.regularization(flow_fields) * self.lambdas["regularization"]
        if self.gan_loss is not None:                                                                # :150-153
            terms["ad_gen"] = self.gan_loss(generated, target)
        if self.style_content_loss is not None:                                                      # :168-176
            terms["style_content_gen"] = self.style_content_loss(generated, target)
        total = sum(terms.values())
        total.backward()          # the reducer's hooks launch each bucket's all-reduce as soon as it is complete
        self.reducer.finish()
        self.optimizer_G.step()
        self.losses = {k: float(v.detach()) for k, v in terms.items()}
        return self.losses
