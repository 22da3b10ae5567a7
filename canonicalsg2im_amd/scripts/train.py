"""Command-line trainer for the HIP hot path on SYNTHETIC batches.

The reference's scripts/train.py owns data loading, logging, evaluation and checkpoint policy; of it only the
iteration (:353-393, :468-485) is on the hot path and lives in `canonicalsg2im_amd.train.Trainer`.  This entry point
drives that iteration with the reference's flags on seeded synthetic batches of the chosen dataset's shape (real
datasets are out of scope), one process per GPU:

    python -m canonicalsg2im_amd.scripts.train --dataset packed_clevr --image_size 256,256 --batch_size 48 \\
        --num_iterations 100 --no_vgg_loss --learned_transitivity 1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m canonicalsg2im_amd.scripts.train ...

For packed datasets the scene graphs are built on the device from the boxes (`sg2im.data.canonical_triplets`), as
the packed data loaders do on the host."""
import os
import sys
import time

import torch


def _vocab_kind(dataset):
    return {"vg": "vg", "packed_vg": "vg", "clevr": "clevr", "packed_clevr": "clevr"}.get(dataset, "coco")


def main(argv=None):
    from .. import dist as csg_dist, train as T
    from ..sg2im.data import canonical_triplets
    from ..synth import BatchConfig, make_batch, make_vocab
    from .args import build_parser, init_args
    args = build_parser().parse_args(argv)
    rank, world, local = csg_dist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("canonicalsg2im_amd needs a HIP device: there is no CPU path")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    args.vocab = make_vocab(_vocab_kind(args.dataset))
    if world > 1:
        args.gpu_ids = ",".join(str(i) for i in range(world))
    init_args(args)
    per_rank = args.batch_size // max(world, 1)
    torch.manual_seed(0)
    trainer = T.Trainer(args, dev)
    t0 = epoch = 0
    if args.restore_checkpoint:
        # scripts/train.py:29-60: `--checkpoint_name` is the PATH of the checkpoint file; a failed restore raises
        if not args.checkpoint_name or not os.path.isfile(args.checkpoint_name):
            raise NotImplementedError("Could not restore weights for checkpoint %s because `no such file` (pass the path "
                                      "of a checkpoint, e.g. <output_dir>/itr_<t>.pt)" % args.checkpoint_name)
        t0, epoch = trainer.load_checkpoint(args.checkpoint_name)
    packed = args.dataset.startswith("packed")
    lo = args.min_objects or (16 if packed else 3)
    hi = args.max_objects or (40 if packed else 8)
    cfg = BatchConfig(per_rank, args.image_size[0], lo, hi, "packed" if packed else "random", mask_size=args.mask_size)
    tic = time.time()
    for t in range(t0 + 1, args.num_iterations + 1):
        batch = [None if x is None else x.to(dev) for x in make_batch(args.vocab, cfg, seed=t * max(world, 1) + rank)]
        if packed:                       # canonical graph from the geometry, on the device
            objs, boxes = batch[1], batch[2]
            n = (objs[..., 0] != 0).sum(1) + 1                 # real objects + the __image__ row appended below
            O = objs.shape[1] + 1
            objs = torch.cat([objs, objs.new_zeros(objs.shape[0], 1, objs.shape[2])], 1)
            boxes = torch.cat([boxes, boxes.new_full((boxes.shape[0], 1, 4), -1.0)], 1)
            centers = boxes[..., :2] + 0.5 * boxes[..., 2:]
            batch[1], batch[2] = objs, boxes
            conv_w = None
            if args.learned_converse:    # the data loader reads the model's converse weights back (scripts/train.py:274-276)
                from ..sg2im.model import get_conv_converse
                conv_w = get_conv_converse(trainer.model).detach().cpu().numpy()
            batch[3], batch[4], batch[5] = canonical_triplets(objs, boxes, centers, n, args.vocab,
                                                              learned_transitivity=bool(args.learned_transitivity),
                                                              learned_converse=bool(args.learned_converse),
                                                              converse_weights=conv_w)
            assert batch[3].shape[1] > 0 and objs.shape[1] == O
        G, D = trainer.step(batch)
        if rank == 0 and (t % args.print_every == 0 or t == args.num_iterations):
            torch.cuda.synchronize()
            rate = args.print_every * args.batch_size / max(time.time() - tic, 1e-9)
            tic = time.time()
            terms = " ".join("%s %.4f" % (k, float(v.detach())) for k, v in list(G.items()) + list(D.items()) if v.numel() == 1)
            print("t = %d / %d  [%.1f img/s]  %s" % (t, args.num_iterations, rate, terms), flush=True)
        if args.output_dir and t % args.checkpoint_every == 0:
            os.makedirs(args.output_dir, exist_ok=True)
            trainer.save_checkpoint(os.path.join(args.output_dir, "itr_%s.pt" % t), t, epoch)      # scripts/train.py:427
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
