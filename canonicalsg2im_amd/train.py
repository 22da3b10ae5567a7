"""One training iteration of the reference trainer (scripts/train.py:346-393, 468-485) on the HIP
modules, data-parallel over processes.

`Trainer.step(batch)` reproduces the reference's order of operations:
  model forward (graph encoder; generator on GT boxes) -> generator losses (2 D passes) ->
  zero_grad / backward / Adam step -> discriminator losses on the detached image (2 D passes) ->
  per-discriminator zero_grad / backward / Adam step.
Differences that are unobservable in the results: during the generator
backward the discriminator's parameters do not require grad, so the weight gradients the reference
computes and then discards are never computed; errors are raised, not swallowed (§9 item 14)."""
import torch

from . import dist as csg_dist
from . import graphs as csg_graphs
from .scripts.args import make_opt  # noqa: F401  (re-exported)
from .scripts.graphs_utils import calc_log_p
from .sg2im.model import get_conv_converse
from .sg2im.meta_models import MetaDiscriminatorModel, MetaGeneratorModel
from .sg2im.pix2pix_model import Pix2PixModel


class Trainer:
    def __init__(self, opt, device):
        self.opt, self.device = opt, device
        self.model = MetaGeneratorModel(opt, device)
        # HIP-graph replay of the shape-static part of the step (graphs.py): single process, default objective
        self.d_frozen = False
        use_graphs = csg_graphs.StepGraphs.supported(self)
        # (the image discriminator's Adam step is captured: its step counter has to live on the device)
        self.discriminator = MetaDiscriminatorModel(opt).to(device).build_optimizers(opt, capturable_img=use_graphs)
        self.gans_model = Pix2PixModel(opt, discriminator=self.discriminator).to(device)
        self.model.train()
        if getattr(opt, "freeze", 0):
            self.freeze_weights(opt.freeze_options)
        csg_dist.broadcast_module(self.model)
        csg_dist.broadcast_module(self.discriminator)
        # param groups of scripts/train.py:312-322
        converse = ['sg_to_layout.module.converse_candidates_weights']
        trans = ['sg_to_layout.module.trans_candidates_weights']
        named = list(self.model.named_parameters())
        base = [p for n, p in named if n not in converse + trans]
        fused = {'fused': True} if torch.device(device).type == 'cuda' else {}   # one multi-tensor kernel per step
        self.optimizer = torch.optim.Adam([{'params': base, 'lr': opt.learning_rate},
                                           {'params': [p for n, p in named if n in trans], 'lr': 1e-2}], **fused)
        self.converse_params = [p for n, p in named if n in converse]
        self.optimizer_converse = torch.optim.Adam([{'params': self.converse_params, 'lr': 1e-2}])
        meta = [opt.vocab['pred_name_to_idx'][p] for p in ("__padding__", "__in_image__")]          # train.py:343-345
        self.non_meta_relations = sorted(set(opt.vocab['pred_name_to_idx'].values()) - set(meta))
        self.converse_buckets = csg_dist.GradBuckets(self.converse_params)
        self.g_buckets = csg_dist.GradBuckets(base + [p for n, p in named if n in trans])
        self.d_params = list(self.discriminator.img_discriminator.parameters())
        self.d_buckets = csg_dist.GradBuckets(self.d_params)
        self.dobj_params, self.dobj_buckets = [], None
        if not opt.use_img_disc:
            csg_dist.broadcast_module(self.discriminator.obj_discriminator)
            self.dobj_params = list(self.discriminator.obj_discriminator.parameters())
            self.dobj_buckets = csg_dist.GradBuckets(self.dobj_params)
        self.dmask_params, self.dmask_buckets = [], None
        if not opt.use_img_disc and opt.mask_size > 0:
            csg_dist.broadcast_module(self.discriminator.mask_discriminator)
            self.dmask_params = list(self.discriminator.mask_discriminator.parameters())
            self.dmask_buckets = csg_dist.GradBuckets(self.dmask_params)
        self.sg_params = [p for p in self.model.sg_to_layout.parameters()] if self.model.has_graph else []
        self.g_params = [p for p in self.model.layout_to_image_model.parameters()] if self.model.has_image else []
        self._grads_dirty, self._eager_steps = False, 0
        self.graphs = csg_graphs.StepGraphs(self) if use_graphs else None
        self.bucket_generation = ()            # (N > 1) generations of the gradient buckets, refreshed at the top of every step
        self.use_graphs = True                 # False: run eagerly without dropping the captured graphs (bench.py's event legs)

    def freeze_weights(self, module):
        """`--freeze 1 --freeze_options generation` (scripts/train.py:104-117, 337-338): the layout-to-image model and
        every discriminator stop training; only the graph encoder is updated."""
        if module != 'generation':
            raise NotImplementedError('Unrecognized option, you can freeze either graph module or I3D module')
        if hasattr(self.model, 'layout_to_image_model'):
            for p in self.model.layout_to_image_model.parameters():
                p.requires_grad = False
        for p in self.discriminator.parameters():
            p.requires_grad = False
        self.d_frozen = True

    def _d_requires_grad(self, flag):
        if self.d_frozen:
            return
        for p in self.d_params + self.dobj_params + self.dmask_params:
            p.requires_grad_(flag)

    def _converse_step(self, r, conv_counts):
        """REINFORCE update of `converse_candidates_weights` (scripts/train.py:370-381): the per-sample box loss,
        normalised over the GLOBAL batch, weights the log-probability of the converse edges the data loader
        sampled for that sample (`conv_counts`).  The data loader reads the updated weights back on the host."""
        eps = float(torch.finfo(torch.float32).eps)                              # np.finfo(np.float32).eps (:345)
        r_all = r
        if csg_dist.active():
            parts = [torch.empty_like(r) for _ in range(csg_dist.world_size())]
            torch.distributed.all_gather(parts, r.contiguous())
            csg_dist.comm_note("allgather", r.numel() * 4 * csg_dist.world_size())
            r_all = torch.cat(parts)
        if r_all.numel() > 1:
            r = (r - r_all.mean()) / (r_all.std() + eps)
        log_prob = calc_log_p(get_conv_converse(self.model), self.non_meta_relations, conv_counts)
        loss_conv = torch.mean(r * log_prob)
        self.optimizer_converse.zero_grad(set_to_none=True)
        self.converse_buckets.begin()
        loss_conv.backward()
        self.converse_buckets.all_reduce_mean()
        self.optimizer_converse.step()
        return loss_conv.detach()

    def step(self, batch):
        """One training iteration.  Shape keys that repeat are replayed from captured HIP graphs (graphs.py); anything
        else — a new shape, N > 1 ranks, masks, `--learned_converse` — runs the eager path below, in this process."""
        if csg_dist.active():
            # a late-gradient flag of the previous iteration re-agrees the bucket set HERE, on every rank and on either path
            # (dist.GradBuckets.resolve): a rebuild re-allocates the flats that captured graphs address
            self.bucket_generation = tuple(b.resolve() for b in (self.g_buckets, self.d_buckets, self.dobj_buckets,
                                                                 self.dmask_buckets) if b is not None)
        if self.graphs is not None and self.use_graphs:
            out = self.graphs.step(batch)
            if out is not None:
                return out
        self._eager_steps += 1
        if self.graphs is not None and not self._grads_dirty:
            # first eager iteration after replayed ones: the modules still hold tensors of the captured autograd graph
            csg_graphs._drop_stale_autograd(self.model, self.discriminator)
        self._grads_dirty = True
        return self._step_eager(batch)

    def _step_eager(self, batch):
        opt = self.opt
        imgs, objs, boxes, triplets, conv_counts, triplet_type, masks, image_ids = batch
        if not opt.use_img_disc and objs.is_cuda:
            self.discriminator.obj_discriminator.prefetch_index(objs)
        model_out = self.model(objs, triplets, triplet_type, boxes_gt=boxes, masks_gt=masks, test_mode=False)
        self.last_model_out = tuple(None if t is None else t.detach() for t in model_out)
        # ---- generator update (train.py:361-368)
        self._d_requires_grad(False)
        G = self.gans_model(batch, model_out, mode="compute_generator_loss")
        G = {k: (v if k == "bbox_pred_all" else v.mean()) for k, v in G.items()}
        self.optimizer.zero_grad(set_to_none=True)
        # N > 1 with graph replay available: an eager step must issue its collectives in the order a REPLAYED step does
        # (graphs.py: every bucket's all-reduce after the backward, in bucket order) — two ranks may take different paths in
        # the same iteration (a shape key one of them has not captured yet), and RCCL matches collectives by issue order.
        # Without replay the hooks launch each bucket as soon as it is complete, under the rest of the backward.
        in_hooks = self.graphs is None
        self.g_buckets.begin(launch=in_hooks)
        G["total_loss"].backward()
        # N > 1: each 64 MB bucket of the generator's gradients (~375 MB in all) is all-reduced on RCCL's stream as
        # soon as the backward has filled it, and the tail keeps travelling while the discriminator losses below are
        # computed — they read neither the generator's parameters nor its gradients (imgs_pred is detached), so
        # applying the generator's Adam step after them changes nothing.
        g_pending = csg_dist.active()
        if g_pending:
            self.g_buckets.flush()
        else:
            self.optimizer.step()
        self._d_requires_grad(True)
        if opt.learned_converse:
            G["loss_conv"] = self._converse_step(G["bbox_pred_all"].detach(), conv_counts)
        # ---- discriminator update (train.py:388-393, 468-472)
        D = {}
        if not opt.skip_generation and opt.freeze_options != "generation":
            D = self.gans_model(batch, model_out, mode="compute_discriminator_loss")
            D = {k: v.mean() for k, v in D.items()}
            # the three discriminators' backward passes are independent: every all-reduce is in flight (asynchronous)
            # before the first optimiser step waits for its own
            self.discriminator.optimizer_d_img.zero_grad(set_to_none=True)
            self.d_buckets.begin(launch=in_hooks)
            D["total_img_loss"].backward()
            self.d_buckets.flush()
            if not opt.use_img_disc:                                    # train.py:478-480
                self.discriminator.optimizer_d_obj.zero_grad(set_to_none=True)
                self.dobj_buckets.begin()
                D["total_obj_loss"].backward()
                self.dobj_buckets.flush()
            do_mask = opt.mask_size > 0 and "total_mask_loss" in D       # train.py:482-485
            if do_mask:
                self.discriminator.optimizer_d_mask.zero_grad(set_to_none=True)
                self.dmask_buckets.begin()
                D["total_mask_loss"].backward()
                self.dmask_buckets.flush()
            self.d_buckets.finish()
            self.discriminator.optimizer_d_img.step()
            if not opt.use_img_disc:
                self.dobj_buckets.finish()
                self.discriminator.optimizer_d_obj.step()
            if do_mask:
                self.dmask_buckets.finish()
                self.discriminator.optimizer_d_mask.step()
        if g_pending:
            self.g_buckets.finish()
            self.optimizer.step()
        if not opt.use_img_disc:
            self.discriminator.obj_discriminator.release_index()        # the prefetched object list dies with its batch
        # values only: a caller that keeps the dictionaries must not keep the step's autograd graph (and with it every
        # parameter's AccumulateGrad node, bound to the stream it was created on) alive into the next iteration
        return {k: v.detach() for k, v in G.items()}, {k: v.detach() for k, v in D.items()}


    # ------------------------------------------------------------------ checkpoints (scripts/train.py:488-520)
    def checkpoint_dict(self, t=0, epoch=0):
        """The reference's checkpoint dictionary: same top-level keys, same state_dict keys inside (its
        `gans_model` is wrapped in DataParallel, hence the `module.` prefix of `gans_model_state`)."""
        d, opt = self.discriminator, self.opt
        out = {
            'model_state': self.model.state_dict(),
            'gans_model_state': {'module.' + k: v for k, v in self.gans_model.state_dict().items()},
            'd_img_state': d.img_discriminator.state_dict(),
            'd_img_optim_state': d.optimizer_d_img.state_dict(),
            'optim_state': self.optimizer.state_dict(),
            'vocab': opt.vocab,
            'counters': {'t': t, 'epoch': epoch},
        }
        if not opt.use_img_disc:
            out.update({'d_obj_state': d.obj_discriminator.state_dict(), 'd_mask_state': d.mask_discriminator.state_dict(),
                        'd_obj_optim_state': d.optimizer_d_obj.state_dict(),
                        'd_mask_optim_state': d.optimizer_d_mask.state_dict()})
        return out

    def save_checkpoint(self, path, t=0, epoch=0):
        if csg_dist.rank() == 0:
            torch.save(self.checkpoint_dict(t, epoch), path)

    def load_checkpoint(self, ckpt, optimizers=True):
        """Restore from a checkpoint dictionary (or a path to one) written by this trainer or by the reference's
        `save_checkpoint`.  Returns (t, epoch)."""
        if not isinstance(ckpt, dict):
            ckpt = torch.load(ckpt, map_location=self.device)
        d = self.discriminator
        # strict, as the reference's restore_checkpoint (scripts/train.py:40-43): a checkpoint written with other
        # flags (--skip_graph_model, --mask_size, num_upsampling_layers ...) must not load partially
        self.model.load_state_dict(ckpt['model_state'])
        d.img_discriminator.load_state_dict(ckpt['d_img_state'])
        if not self.opt.use_img_disc:
            d.obj_discriminator.load_state_dict(ckpt['d_obj_state'])
            if 'd_mask_state' in ckpt:                  # absent from the reference's own save_checkpoint (:488-520)
                d.mask_discriminator.load_state_dict(ckpt['d_mask_state'])
        if optimizers:
            _load_optimizer(self.optimizer, ckpt['optim_state'])
            _load_optimizer(d.optimizer_d_img, ckpt['d_img_optim_state'])
            if not self.opt.use_img_disc and 'd_obj_optim_state' in ckpt:
                _load_optimizer(d.optimizer_d_obj, ckpt['d_obj_optim_state'])
                if 'd_mask_optim_state' in ckpt:
                    _load_optimizer(d.optimizer_d_mask, ckpt['d_mask_optim_state'])
        from . import ops
        ops.invalidate_weight_caches()
        if self.graphs is not None:
            # optimizer.load_state_dict REPLACES the moment tensors the captured Adam step updates in place: the graphs
            # would keep stepping the old ones.  Drop them; the next repeated shape is captured afresh.
            self.graphs.invalidate()
            self._grads_dirty = True
        c = ckpt.get('counters', {})
        return c.get('t', 0), c.get('epoch', 0)


def _load_optimizer(optimizer, state):
    """`optimizer.load_state_dict(state)` keeping THIS optimiser's implementation flags.  `load_state_dict` takes every
    entry of the saved param_groups — learning rate and betas (wanted: the reference restores them the same way,
    scripts/train.py:40-60), but also `fused` / `capturable` / `foreach`, and it leaves the step counters on the host when
    the saved optimiser was not fused.  A checkpoint written by the reference (plain Adam) would so turn the one-kernel
    fused step off and make the image discriminator's step un-capturable."""
    keep = [{k: g[k] for k in ('fused', 'capturable', 'foreach') if k in g} for g in optimizer.param_groups]
    optimizer.load_state_dict(state)
    for g, flags in zip(optimizer.param_groups, keep):
        g.update(flags)
        if g.get('fused') or g.get('capturable'):
            for p in g['params']:
                st = optimizer.state.get(p)
                if st and 'step' in st:
                    st['step'] = torch.as_tensor(st['step'], dtype=torch.float32).to(p.device)


# ------------------------------------------------------------------ test / smoke helpers
def split_state(trainer):
    """(sg, g, d) state dicts with the reference's un-prefixed keys."""
    sg = {k: v for k, v in trainer.model.sg_to_layout.module.state_dict().items()} \
        if hasattr(trainer.model, "sg_to_layout") else {}
    g = {k: v for k, v in trainer.model.layout_to_image_model.module.state_dict().items()} \
        if hasattr(trainer.model, "layout_to_image_model") else {}
    d = dict(trainer.discriminator.img_discriminator.state_dict())
    return sg, g, d


def obj_disc_state(trainer):
    return dict(trainer.discriminator.obj_discriminator.state_dict())


def state_snapshot(trainer):
    """CPU copies of everything `oracle_state_from` needs (weights, buffers, opt) as plain dicts — taken BEFORE a
    step so that a checker can replay the same step later (bench.py's `parity_b16`)."""
    sg, g, d = split_state(trainer)
    cpu = lambda sd: None if sd is None else {k: v.detach().cpu().clone() for k, v in sd.items()}
    snap = {"opt": trainer.opt, "sg": cpu(sg), "g": cpu(g), "d": cpu(d), "dobj": None, "vgg": None, "dmask": None,
            "noise": None}
    if not trainer.opt.use_img_disc:
        snap["dobj"] = cpu(obj_disc_state(trainer))
    if hasattr(trainer.gans_model, "criterionVGG"):
        snap["vgg"] = cpu(trainer.gans_model.criterionVGG.vgg.state_dict())
    if not trainer.opt.use_img_disc and trainer.opt.mask_size > 0:
        snap["dmask"] = cpu(dict(trainer.discriminator.mask_discriminator.state_dict()))
        noise = trainer.model.sg_to_layout.module.mask_noise
        snap["noise"] = None if noise is None else noise.detach().cpu().clone()
    return snap


def oracle_state_from(trainer, oracle_mod):
    """CPU copy of the trainer's weights (or of a `state_snapshot` of them) as an `oracle.TrainState` (used by tests,
    smoke() and bench.py's checker leg only; the oracle module is passed in so that this package never imports it)."""
    snap = trainer if isinstance(trainer, dict) else state_snapshot(trainer)

    def leafs(sd, skip=()):
        out = {}
        for k, v in sd.items():
            if any(s in k for s in skip):
                continue
            t = v.detach().cpu().clone()
            is_buf = any(s in k for s in ("running_", "weight_u", "weight_v", "num_batches_tracked"))
            if t.is_floating_point() and not is_buf:
                t.requires_grad_(True)
            out[k] = t
        return out
    sg = leafs(snap["sg"])
    if "trans_candidates_weights" in sg:
        for k in list(sg):
            if k.endswith("predicates_transitive_weights"):
                sg[k] = sg["trans_candidates_weights"]
    unused = ("repr_net", "image_encoder")
    dobj = leafs(snap["dobj"]) if snap["dobj"] is not None else None
    vgg = snap["vgg"]
    dmask = leafs(snap["dmask"]) if snap["dmask"] is not None else None
    return oracle_mod.TrainState(snap["opt"], sg, leafs(snap["g"], unused), leafs(snap["d"], unused), dobj, vgg, dmask,
                                 snap["noise"])
