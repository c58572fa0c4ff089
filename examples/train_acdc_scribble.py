#!/usr/bin/env python3
"""End-to-end trainer on the reference's ACDC layout, built only from this package's pieces -- the flow of
code/train_weakly_supervised_pCE_WSL4MIS (ours_proposed) / ..._pCE_GatedCRFLoss_2D.py, one process per GPU:

    BaseDataSets(h5lite) -> BatchRandomGenerator (device augmentation) -> TrainEngine.step -> val_2D metrics

    python examples/train_acdc_scribble.py --root_path <.../data/ACDC> --fold fold1 --max_iterations 60000
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_acdc_scribble.py ...

Each rank draws its own batches (independent shuffles), gradients are averaged by the engine."""
import argparse
import os
import random
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before HIP loads: keeps RCCL's streams off the decoder side stream's queue (bench.py)

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsl4mis_amd import val_2D  # noqa: E402
from wsl4mis_amd.dataloaders.dataset import BaseDataSets, BatchRandomGenerator  # noqa: E402
from wsl4mis_amd.engine import TrainEngine  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--root_path", required=True)
    ap.add_argument("--fold", default="fold1")
    ap.add_argument("--sup_type", default="scribble")
    ap.add_argument("--model", default="unet_cct", choices=["unet_cct", "unet"])
    ap.add_argument("--loss", default="ours_proposed", choices=["ours_proposed", "pce", "pce_gatedcrf", "pce_tv", "pce_ms", "pce_entropy", "ce_dice", "mean_teacher", "ustm"])
    ap.add_argument("--num_classes", type=int, default=4)
    ap.add_argument("--max_iterations", type=int, default=60000)
    ap.add_argument("--stop_iterations", type=int, default=0, help="stop after this many iterations while keeping the poly schedule "
                    "of --max_iterations (short-schedule comparisons against the oracle: tests/acdc_oracle_arm/oracle_acdc_short.py)")
    ap.add_argument("--batch_size", type=int, default=12)
    ap.add_argument("--base_lr", type=float, default=0.01)
    ap.add_argument("--patch_size", type=int, nargs=2, default=[256, 256])
    ap.add_argument("--seed", type=int, default=2022)
    ap.add_argument("--val_every", type=int, default=200)
    ap.add_argument("--labeled_type", default="all", help="'all' = every training patient of the fold (upstream WSL4MIS's weakly-"
                    "supervised runs), 'labeled' / 'unlabeled' = the subsets of dataset_semi.py")
    ap.add_argument("--snapshot_path", default=None, help="write the reference's checkpoints here (state_dict .pth files)")
    ap.add_argument("--save_every", type=int, default=3000)
    ap.add_argument("--no_hd95", action="store_true", help="validation: Dice only (the model-selection metric), skip HD95")
    ap.add_argument("--log_every", type=int, default=20)
    ap.add_argument("--quiet", action="store_true", help="no per-iteration lines (validation lines stay)")
    ap.add_argument("--curve_json", default=None, help="write the loss / validation curve here (rank 0)")
    ap.add_argument("--oracle_stream", action="store_true", help="draw the nn.Dropout masks on the CPU generator in the order "
                    "tests/acdc_oracle_arm/oracle_acdc_short.py draws them (with --resume of its initial state and the same --seed, the two arms then "
                    "run the SAME trajectory up to fp32 round-off: profiles/r3_acdc_short_schedule.md)")
    ap.add_argument("--resume", default=None, help="a state_dict .pth (the reference's or ours: same keys) to start from")
    args = ap.parse_args(argv)

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    random.seed(args.seed), np.random.seed(args.seed + rank), torch.manual_seed(args.seed)
    train = BaseDataSets(base_dir=args.root_path, split="train", fold=args.fold, sup_type=args.sup_type,
                         labeled_type=args.labeled_type, cache=True)
    val = BaseDataSets(base_dir=args.root_path, split="val", fold=args.fold, cache=True)
    if len(train) == 0:
        raise SystemExit("no training slices for this fold under " + args.root_path)
    aug = BatchRandomGenerator(args.patch_size, device_cache=True)
    eng = TrainEngine(args.model, 1, args.num_classes, base_lr=args.base_lr, max_iterations=args.max_iterations,
                      loss=args.loss)
    if args.resume:
        eng.model.load_state_dict(torch.load(args.resume, map_location="cpu"))
    if args.snapshot_path and rank == 0:
        os.makedirs(args.snapshot_path, exist_ok=True)
    torch.manual_seed(args.seed + 1000 * rank)          # dropout masks differ per rank; beta is shared (python RNG)
    order = np.random.RandomState(args.seed + rank)
    it, best, history = 0, 0.0, []
    log = [] if (args.curve_json and rank == 0) else None
    import time
    t_start = time.time()
    last = args.stop_iterations if 0 < args.stop_iterations < args.max_iterations else args.max_iterations
    while it < last:
        perm = order.permutation(len(train))
        for b in range(0, len(perm), args.batch_size):
            idx = perm[b:b + args.batch_size]
            if len(idx) < 2:                            # BatchNorm needs more than one slice
                continue
            image, label = aug([train[int(i)] for i in idx])
            if args.oracle_stream:
                from wsl4mis_amd.networks.unet import _DROP, _FT
                n, (ph, pw) = len(idx), args.patch_size
                em = [(torch.rand((n, _FT[l], ph >> l, pw >> l)) >= _DROP[l]).to(torch.uint8).cuda() for l in range(5)]
                eng.model.set_dropout_masks(em, None)
            eng.step(image, label, random.random() + 1e-10)
            if args.oracle_stream:
                eng.model.set_dropout_masks(None, None)
            it += 1
            if rank == 0 and (it % args.log_every == 0 or it == 1):
                o = eng.losses()
                history.append((it, o["loss"]))
                if log is not None:
                    log.append(dict(o, iteration=it, lr=eng.lr))
                if not args.quiet:
                    print("iteration %d : " % it + ", ".join(f"{k} {v:.4f}" for k, v in o.items()), flush=True)
            if len(val) and it % args.val_every == 0:
                # every rank validates its share of the volumes, then one all-reduce of the sums: nobody sits blocked in the next
                # step's gradient all-reduce while rank 0 works through the whole set.  Parameters are bit-identical on all ranks
                # (all-reduced gradients); the BatchNorm running statistics are NOT -- every rank saw other batches -- so rank 0's
                # buffers are broadcast first (what DistributedDataParallel(broadcast_buffers=True) does before every forward):
                # the Dice that is reported and written into the checkpoint's file name then belongs to the state_dict rank 0
                # saves
                if world > 1:
                    dist.broadcast(eng.model._buf_arena, src=0)
                    dist.broadcast(eng.model._nbt, src=0)
                volume_fn = val_2D.test_single_volume_cct if args.model == "unet_cct" else val_2D.test_single_volume
                acc = np.zeros(2 * (args.num_classes - 1) + 1)
                for i in range(rank, len(val), world):
                    v = val[i]
                    m = np.array(volume_fn(v["image"], v["label"], eng.model, args.num_classes, args.patch_size,
                                           with_hd95=not args.no_hd95), dtype=np.float64)
                    acc += np.concatenate([m[:, 0], m[:, 1], [1.0]])
                if world > 1:
                    t = torch.from_numpy(acc).cuda()
                    dist.all_reduce(t)
                    acc = t.cpu().numpy()
                nc = args.num_classes - 1
                dice_c, hd_c = acc[:nc] / acc[-1], acc[nc:2 * nc] / acc[-1]
                mean_dice, mean_hd95 = float(dice_c.mean()), float(hd_c.mean())
                if rank == 0:
                    if mean_dice > best and args.snapshot_path:      # ..._ours_proposed.py:174-182
                        for name in ("iter_{}_dice_{}.pth".format(it, round(mean_dice, 4)), "{}_best_model.pth".format(args.model)):
                            torch.save(eng.model.state_dict(), os.path.join(args.snapshot_path, name))
                    print("iteration %d : mean_dice %.4f mean_hd95 %.3f (best %.4f)" % (it, mean_dice, mean_hd95, max(best, mean_dice)),
                          flush=True)
                    if log is not None:
                        log.append({"iteration": it, "mean_dice": mean_dice, "mean_hd95": mean_hd95,
                                    "dice_per_class": [float(x) for x in dice_c]})
                best = max(best, mean_dice)
                eng.model.train()
            if rank == 0 and args.snapshot_path and it % args.save_every == 0:     # ..._ours_proposed.py:194-198
                torch.save(eng.model.state_dict(), os.path.join(args.snapshot_path, "iter_" + str(it) + ".pth"))
            if it >= last:
                break
    if log is not None:
        import json
        with open(args.curve_json, "w") as fh:
            json.dump({"args": vars(args), "train_slices": len(train), "val_volumes": len(val), "best_mean_dice": best,
                       "wall_seconds": round(time.time() - t_start, 1), "world": world, "curve": log}, fh)
    return history


if __name__ == "__main__":
    main()
