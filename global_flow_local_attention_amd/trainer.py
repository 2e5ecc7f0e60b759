"""One-process-per-GPU training shell around the UNMODIFIED reference generators.

The reference trains through `BaseModel` (model/base_model.py) + `nn.DataParallel` (model/face_model.py:81-93):
one Python process scatters the batch over the GPUs, gathers outputs and reduces gradients through GPU 0.  Here every
GPU has its own process (torch.distributed over RCCL), owns a slice of the batch, and the gradients are averaged by a
bucketed all-reduce launched from autograd hooks (dist.GradBucketReducer), overlapping the rest of backward.

* `load_reference_checkpoint(net, path)` -- `BaseModel.load_networks` semantics (base_model.py:154-197): exact load,
  else keys present in the network (optionally with / without the `module.` prefix DataParallel checkpoints carry),
  else shape-matching entries only; returns what was not initialised.
* `TrainerShell` -- one `optimize_parameters` step (pose_model.py:186-196) for one rank: shard, forward, loss terms,
  backward (reducer hooks), optimizer step.  The adversarial and VGG style/content terms need the reference's
  discriminators / a pretrained VGG19 and are injected as callables (stubbed to zero by default, as BASELINE config 4
  prescribes); the L1 reconstruction, sampling-correctness and affine-regularisation terms use this package's ops.
"""
import torch
import torch.nn as nn

from . import dist as gdist


def load_reference_checkpoint(net, path_or_state, map_location="cpu"):
    """Load a reference `<epoch>_net_G.pth` into `net` the way BaseModel.load_networks does (base_model.py:154-197).
    Returns the sorted list of top-level submodules that were NOT initialised from the checkpoint (empty = exact)."""
    state = torch.load(path_or_state, map_location=map_location) if isinstance(path_or_state, str) else path_or_state
    try:
        net.load_state_dict(state)
        return []
    except RuntimeError:
        pass
    model_dict = net.state_dict()
    picked = {k: v for k, v in state.items() if k in model_dict}
    if not picked:  # checkpoint written from a DataParallel wrapper, or the other way round
        picked = {k.replace("module.", ""): v for k, v in state.items() if k.replace("module.", "") in model_dict}
    if not picked:
        picked = {("module." + k): v for k, v in state.items() if "module." + k in model_dict}
    try:
        net.load_state_dict(picked)  # checkpoint has excessive layers: only the used ones
        return []
    except RuntimeError:
        pass
    not_initialized = set()
    for k, v in picked.items():
        if v.size() == model_dict[k].size():
            model_dict[k] = v
    for k, v in model_dict.items():
        if k not in picked or v.size() != picked[k].size():
            not_initialized.add(k.split(".")[0])
    net.load_state_dict(model_dict)
    return sorted(not_initialized)


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
