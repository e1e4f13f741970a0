"""Drop-in CLI for the reference entry point (main.py:151-235):

    python3 vip-cup-2022_amd/main.py <input.csv> <output.csv> [--scores-out scores.csv] [--synthetic]
    python -m torch.distributed.run --nproc-per-node N ... vip-cup-2022_amd/main.py in.csv out.csv

Same contract: the input CSV has a ``filename`` column with paths relative to the CSV's directory (main.py:77-79,
155-164); the output CSV has columns ``filename,logit`` with logit in {0.0, 1.0} = (ensemble mean > 0.487)
(main.py:143-145,225).  ``--scores-out`` additionally writes the continuous ensemble mean (the reference keeps
it only in memory).  The ensemble manifest is ``ckpts/ckpts.json`` ([name, [H,W], idx],
main.py:171-198); members whose graph is not built yet are reported and skipped only under ``--allow-missing``.

Checkpoints: ``<script dir>/ckpts/<name>/ckpt/*.h5`` (Keras weight / model files, as in the reference), else ``ckpt/saved_model.pb`` (a
Keras SavedModel directory: its variables are read by ``tfbundle``), or ``*.npz`` (a flat dict of
Keras-named arrays), one file per fold.
The reference ships none (README.md:13) and raises when a directory is empty (main.py:194); so does this CLI,
unless ``--synthetic`` asks for the seeded synthetic checkpoints.
"""
import argparse
import json
import os
import sys
import time
from glob import glob

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("input_csv")
    ap.add_argument("output_csv")
    ap.add_argument("--scores-out", default=None)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--allow-missing", action="store_true")
    ap.add_argument("--ckpt-cfg", default=os.path.join(HERE, "ckpts", "ckpts.json"))
    ap.add_argument("--batch-size", type=int, default=128)  # main.py:85
    ap.add_argument("--debug", type=int, default=0)         # main.py:82-83: first 100 images
    ap.add_argument("--tta", type=int, default=1)           # main.py:167: passes under apply_augment, averaged
    ap.add_argument("--tta-seed", type=int, default=0)
    ap.add_argument("--shard", default="images", choices=["images", "members", "hybrid"],
                    help="how N > 1 ranks split the (member, image-shard) grid: images = every rank all members on its image shard "
                         "(MirroredStrategy's split, utils/device.py:7); members = rank r owns members r mod N and scores every "
                         "image (one model per GPU); hybrid = LPT packing by measured ms/image.  One all-gather in every mode.")
    ap.add_argument("--precision", default=None, choices=["fast", "strict", "f32"],
                    help="fast (default; env VIP_PRECISION): fp16 storage, the throughput path - member logits at the fp16 storage floor "
                         "(7e-4 ... 8e-3 vs an fp32 run).  strict: fp32 storage and fp32 matrix arithmetic, what the reference's "
                         "TensorFlow run computes in (main.py:107-109): every member's logit within 1e-3 of it, ~3.5x slower.")
    ap.add_argument("--calibration-images", default=None, metavar="DIR",
                    help="fast mode: a directory of JPEGs (up to 32 are read) for the bias calibration of the fp16 weights instead of the "
                         "built-in seeded synthetic batch - use real images with real checkpoints (the correction needs typical "
                         "per-channel input means; inputs only, no labels)")
    ap.add_argument("--no-bias-calibration", action="store_true",
                    help="fast mode: plain fp16 weights, no calibration pass at load time")
    a = ap.parse_args(argv)

    import pandas as pd
    import torch
    import vipcup_amd  # noqa: F401
    from vipcup_amd import ensemble, ops, pipeline, zoo

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("vipcup_amd main: no GPU visible — the HIP path has no CPU fallback")
    backend = os.environ.get("VIP_DIST_BACKEND", "nccl")     # gloo: several ranks may share one card (rehearsals on a 1-GPU box)
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    infer_path = os.path.dirname(os.path.abspath(a.input_csv))   # main.py:161-164
    test_csv = pd.read_csv(a.input_csv)
    names = test_csv.filename.values.tolist()
    if a.debug:
        names = names[:100]
    paths = [os.path.join(infer_path, n) for n in names]

    manifest = []
    for name, dim, idx in json.load(open(a.ckpt_cfg)):           # main.py:171-198
        key = zoo.by_ckpt_name(name)
        if key is None:
            if a.allow_missing:
                if rank == 0:
                    print(f"> SKIP {name}: graph not built in this round")
                continue
            raise ValueError(f"manifest member {name} has no graph in vipcup_amd.zoo")
        manifest.append((name, dim, idx, key))
    mode = a.precision or ops.PRECISION
    calib = None
    if a.calibration_images and mode == "fast" and not a.no_bias_calibration:
        files = sorted(glob(os.path.join(a.calibration_images, "*.jpg")) + glob(os.path.join(a.calibration_images, "*.jpeg")) +
                       glob(os.path.join(a.calibration_images, "*.JPG")))[:32]
        if not files:
            raise ValueError(f"--calibration-images {a.calibration_images}: no *.jpg / *.jpeg files")
        calib = pipeline.decode_jpegs([open(f, "rb").read() for f in files])
        if rank == 0:
            print(f"> BIAS CALIBRATION on {len(files)} images from {a.calibration_images}")
    build = dict(bias_calibration=not a.no_bias_calibration, precision=mode, calibration_batch=calib)
    plan = ensemble.ShardPlan("members" if a.shard == "members" else "images", len(manifest), world)
    mine = {m for ms in plan.units[rank].values() for m in ms}   # members mode: a rank loads only the members it owns
    members = []
    for mi, (name, dim, idx, key) in enumerate(manifest):
        spec = zoo.MEMBERS[key]
        assert [spec.input_hw, spec.input_hw] == list(dim), (name, dim)
        ckpts = sorted(glob(os.path.join(HERE, "ckpts", name, "ckpt", "*.npz")) + glob(os.path.join(HERE, "ckpts", name, "ckpt", "*.h5")))
        sm_path = os.path.join(HERE, "ckpts", name, "ckpt", "saved_model.pb")
        if not ckpts and os.path.isfile(sm_path):                  # SavedModel format, when there are no .h5 folds (main.py:186-191)
            ckpts = [sm_path]
        if mi not in mine:
            if not ckpts and not a.synthetic:
                raise ValueError(f"no checkpoints under ckpts/{name}/ckpt (pass --synthetic for seeded synthetic weights)")
            members.append((spec, None))
            continue
        if ckpts:
            # graph family from the directory name, variant (first_strides / classes / head activation) from the file's model_config
            folds = [zoo.construct(spec, zoo.match_variable_names(spec, zoo.read_checkpoint(c)), variant=zoo.checkpoint_variant(spec, c),
                                   **build) for c in ckpts]
        elif a.synthetic:
            folds = [zoo.build_member(key, **build)[1]]
        else:
            raise ValueError(f"no checkpoints under ckpts/{name}/ckpt (pass --synthetic for seeded synthetic weights)")
        members.append((spec, zoo.FoldMean(folds)))                # mean over folds, main.py:121
        if rank == 0:
            print(f"> MODEL({len(members)}): {name} | DIM: {dim} | folds: {len(folds)} | precision: {mode}")

    def jpegs_for(lo, hi):
        out = []
        for p in paths[lo:hi]:
            with open(p, "rb") as f:
                out.append(f.read())
        return out

    t0 = time.time()
    costs = None
    if a.shard == "hybrid" and world > 1:
        costs = ensemble.measure_costs(members, jpegs_for(0, min(len(paths), a.batch_size)), dist, rank)
        if rank == 0:
            print("> HYBRID PLAN:", ensemble.ShardPlan("hybrid", len(members), world, costs).describe())
    per_model = ensemble.score_files(jpegs_for, len(paths), members, a.batch_size, rank, world, dist,
                                     tta=a.tta, tta_seed=a.tta_seed, shard=a.shard, costs=costs)
    uniq, score, decision = ensemble.aggregate(names, per_model)
    if rank == 0:
        pd.DataFrame({"filename": uniq, "logit": decision}).to_csv(a.output_csv, index=False)  # main.py:143-145
        if a.scores_out:
            cols = {"filename": names, "ensemble_mean": per_model.mean(axis=0)}
            for (spec, _), row in zip(members, per_model):
                cols[spec.name] = row
            pd.DataFrame(cols).to_csv(a.scores_out, index=False)
        dt = time.time() - t0
        print(f"> FINAL PREDICTION SAVED TO {a.output_csv}")
        print(f"> TIME TO INFER: {dt / 60:.2f} min ({len(paths) / dt:.1f} images/s on {world} GPU(s))")  # main.py:231-235
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
