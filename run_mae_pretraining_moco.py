"""Pre-training driver for the MI355X engine: the README command of the reference (README.md:53-78) runs against this file unchanged.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run_mae_pretraining_moco.py \
        --image_alone_path DATA --mask_ratio 0.7 --batch_size 128 --opt adamw --output_dir OUT --epochs 10 --warmup_steps 5000 \
        --max_len 25 --num_view 2 --moco_dim 256 --moco_mlp_dim 4096 --moco_m 0.99 --moco_m_cos --moco_t 0.2 --num_windows 4 \
        --contrast_warmup_steps 0 --contrast_start_epoch 0 --loss_weight_pixel 1. --loss_weight_contrast 0.1 --only_mim_on_ori_img \
        --weight_decay 0.1 --opt_betas 0.9 0.999 --model pretrain_simmim_moco_ori_vit_small_patch4_32x128 \
        --patchnet_name no_patchtrans --encoder_type vit

What it keeps of run_mae_pretraining_moco.py:303-453 (`main`): distributed init from the launcher's environment, per-rank seeding (:313-316),
model factory call with the same keyword set (:280-292), lr = lr * global batch / 256 (:381-382), step-level cosine lr / weight-decay
schedules (:398-406; the reference's quirk that `--warmup_steps` only takes effect while `--warmup_epochs` > 0 included), auto-resume
from `checkpoint-N.pth` (:413), the epoch loop with `train_one_epoch`, `save_model` every `--save_ckpt_freq` epochs and the JSON line
per epoch in `log.txt` (:437-446).

What is different, on purpose: the input pipeline.  The reference decodes LMDB crops and runs imgaug in dataloader workers
(dataset/dataset_image.py -- lmdb, cv2 and imgaug are not part of this image and are out of scope).  Here the data
source is either `--synthetic N` / `--data_path synthetic` (N samples of U(-1,1) crops per epoch: the benchmark's data) or a directory tree
of image files under `--image_alone_path` / `--data_path` (decoded with Pillow to uint8 crops; resize + normalisation + mask drawing happen on the MI355X,
dig_amd/datasets.py).  The augmented view is produced by `--aug_module pkg.fn` (a callable `fn(list_of_uint8_crops) -> list_of_uint8_crops`
run on the host) -- without it the second view is the crop itself.
"""
import argparse
import datetime
import importlib
import json
import os
import random
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")          # before HIP initialises: see dig_amd/__init__.py

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def get_args(argv=None):
    p = argparse.ArgumentParser("DiG pre-training on MI355X (dig_amd)", add_help=True)
    a = p.add_argument
    # recipe (README.md:53-78)
    a("--batch_size", type=int, default=64, help="samples per GPU")
    a("--epochs", type=int, default=300)
    a("--save_ckpt_freq", type=int, default=1)
    a("--eval_freq", type=int, default=500)
    a("--model", type=str, default="pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    a("--encoder_type", type=str, default="vit")
    a("--mask_ratio", type=float, default=0.75)
    a("--num_view", type=int, default=1)
    a("--input_h", type=int, default=32)
    a("--input_w", type=int, default=128)
    a("--drop_path", type=float, default=0.0)
    a("--normlize_target", type=lambda s: str(s).lower() in ("1", "true", "yes"), default=False)
    a("--opt", type=str, default="adamw")
    a("--opt_eps", type=float, default=1e-8)
    a("--opt_betas", type=float, nargs="+", default=None)
    a("--clip_grad", type=float, default=None)
    a("--weight_decay", type=float, default=0.05)
    a("--weight_decay_end", type=float, default=None)
    a("--lr", type=float, default=1.5e-4)
    a("--min_lr", type=float, default=1e-5)
    a("--warmup_epochs", type=int, default=40)
    a("--warmup_steps", type=int, default=-1)
    a("--num_windows", type=int, default=5)
    a("--patchnet_name", type=str, default="regular")
    a("--loss_weight_contrast", type=float, default=0.0)
    a("--contrast_warmup_steps", type=int, default=0)
    a("--contrast_start_epoch", type=int, default=0)
    a("--loss_weight_pixel", type=float, default=1.0)
    a("--only_mim_on_ori_img", action="store_true", default=False)
    a("--queue_size", type=int, default=65536)
    a("--moco_dim", type=int, default=256)
    a("--moco_mlp_dim", type=int, default=4096)
    a("--moco_m", type=float, default=0.99)
    a("--moco_m_cos", action="store_true", help="(dead flag of the reference; the cosine momentum schedule is --use_moco_m_cos, default on)")
    a("--use_moco_m_cos", type=int, default=1)
    a("--moco_t", type=float, default=1.0)
    a("--max_len", type=int, default=25)
    # data
    a("--image_alone_path", nargs="+", type=str, default="", help="directory tree(s) of image files (png / jpg / bmp ...)")
    a("--synthetic", type=int, default=0, help="N > 0: N synthetic samples per epoch instead of --image_alone_path")
    a("--aug_module", type=str, default="", help="pkg.module.fn: host callable producing the augmented crops of the second view")
    a("--num_workers", type=int, default=10)
    # run
    a("--output_dir", type=str, default="")
    a("--log_dir", type=str, default=None)
    a("--device", type=str, default="cuda")
    a("--seed", type=int, default=0)
    a("--resume", type=str, default="")
    a("--auto_resume", action="store_true", default=True)
    a("--no_auto_resume", action="store_false", dest="auto_resume")
    a("--start_epoch", type=int, default=0)
    a("--dist_url", type=str, default="env://")
    a("--data_path", nargs="+", type=str, default="",
      help="`synthetic` (with --synthetic N, default 100 global batches per epoch), or directory tree(s) of image files like --image_alone_path; "
           "an LMDB directory is the reference's dataset/ reader's business (INTEGRATION.md, 'Input transform')")
    a("--num_samples", type=float, default=float("inf"), help="cap on the samples of --data_path taken per epoch")
    a("--aloneimage_num_samples", type=float, default=float("inf"), help="cap on the samples of --image_alone_path taken per epoch")
    a("--use_ema", action="store_true", default=False, help="(the reference's teacher-student mode: not built, raises)")
    # Every other flag of the reference's get_args (run_mae_pretraining_moco.py:30-277), so that its launch scripts run unchanged: the
    # pre-training path of the reference never reads them (leftovers of other experiments: `grep args.<name>` over its driver, engine,
    # optimizer factory, datasets and utils finds nothing), or they configure what has no counterpart here (dataloader pinning, SGD momentum,
    # ModelArts paths, the launcher's rank arguments -- ranks come from the launcher's environment).  Parsed, reported once, not used.
    for name in REFERENCE_ONLY_SWITCHES:
        a(name, action="store_true", default=False, help=argparse.SUPPRESS)
    a("--no_pin_mem", action="store_false", dest="pin_mem", help=argparse.SUPPRESS)
    for name in REFERENCE_ONLY_VALUES:
        a(name, type=str, default=None, help=argparse.SUPPRESS)
    args = p.parse_args(argv)
    given = [n for n in REFERENCE_ONLY_SWITCHES if getattr(args, n[2:])] + [n for n in REFERENCE_ONLY_VALUES if getattr(args, n[2:]) is not None]
    if given:
        print("flags of the reference that its pre-training path does not read either (accepted, not used): " + " ".join(given))
    if args.use_ema:
        raise SystemExit("--use_ema (teacher-student mode, run_mae_pretraining_moco.py:326-338) is not built")
    return args


REFERENCE_ONLY_SWITCHES = ("--alternately_epoch_training", "--alternately_training", "--dist_on_itp", "--first_train_mim", "--fix_mask_token",
                           "--imagenet_default_mean_and_std", "--mix_train_with_aloneimage", "--mix_train_with_ctx", "--only_real_data_for_pretrain",
                           "--pin_mem", "--use_abi_aug", "--use_color_aug", "--use_corner_mask", "--use_hard_sample", "--use_image_slice",
                           "--use_loss_weight", "--use_mem_in_decoder", "--use_mim", "--use_moco", "--use_multiscale_mask", "--use_patch_transformer")
REFERENCE_ONLY_VALUES = ("--attn_map_type", "--aug_ratio", "--cluster_update_interval", "--color_jitter", "--contrast_temperature", "--corner_prob",
                         "--corner_ratio", "--corrupt_ops_ratios", "--ctx_max_len", "--ctx_min_len", "--ctx_nb_classes", "--ctx_num_samples", "--ctx_path",
                         "--distill_start_epoch", "--image_to_ctx_ratio", "--input_size", "--local_rank", "--loss_feat_beta", "--loss_feat_type",
                         "--loss_weight_consist", "--loss_weight_distill", "--loss_weight_feat_align", "--loss_weight_pos", "--loss_weight_semgroup",
                         "--loss_win_size", "--mask_ratios", "--mask_scales", "--momentum", "--momentum_teacher", "--momentum_teacher_end",
                         "--num_distribution", "--num_mem_slots", "--num_relation_heads", "--num_target_layers", "--recon_patch_scales", "--relation_T",
                         "--relation_window_size", "--soft_label_type", "--text_loss_weight", "--text_mask_ratio", "--train_interpolation", "--train_url",
                         "--vis_loss_weight", "--voc_type", "--warmup_lr", "--world_size")


class SyntheticCrops:
    """`n` samples per epoch of U(-1, 1) crops, already normalised and resident on the device (the benchmark's data)."""

    def __init__(self, n, batch, device, seed, mask_gen):
        self.steps, self.batch, self.device, self.mask_gen = n // batch, batch, device, mask_gen
        g = torch.Generator().manual_seed(seed)
        self.pool = [((torch.rand((batch, 3, 32, 128), generator=g) * 2 - 1).to(device), (torch.rand((batch, 3, 32, 128), generator=g) * 2 - 1).to(device))
                     for _ in range(4)]

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            im, au = self.pool[i % len(self.pool)]
            yield ([im, au, self.mask_gen(self.batch)], torch.ones(1), torch.ones(1))


class ImageFolderCrops:
    """Image files under the given roots, sharded over ranks like DistributedSampler(shuffle=True, drop_last) does: every epoch a
    permutation seeded by the epoch number, rank r takes indices r, r + W, ...; decoding on a small thread pool, everything after the
    uint8 crop (bicubic resize to 32 x 128, normalisation, masks) on the device."""
    EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".webp")

    def __init__(self, roots, batch, rank, world, transform, aug, workers, cap=float("inf")):
        self.files = sorted(os.path.join(d, f) for r in roots for d, _, fs in os.walk(r) for f in fs if f.lower().endswith(self.EXT))
        if not self.files:
            raise SystemExit(f"no image files under {roots}")
        if cap < len(self.files):                                        # --num_samples / --aloneimage_num_samples: the first N, as the
            self.files = self.files[:int(cap)]                           # reference's datasets do (dataset/dataset_image.py: min(nSamples, num_samples))
        self.batch, self.rank, self.world, self.transform, self.aug, self.workers = batch, rank, world, transform, aug, max(1, workers)
        self.epoch = 0
        self.steps = len(self.files) // batch // world

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.steps

    @staticmethod
    def _decode(path):
        from PIL import Image
        with Image.open(path) as im:
            return np.asarray(im.convert("RGB"), dtype=np.uint8)

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        order = np.random.RandomState(self.epoch).permutation(len(self.files))[self.rank::self.world]
        with ThreadPoolExecutor(self.workers) as pool:
            pending = None
            for s in range(self.steps + 1):
                nxt = None
                if s < self.steps:
                    idx = order[s * self.batch:(s + 1) * self.batch]
                    nxt = pool.map(self._decode, [self.files[i] for i in idx])          # decode step s while step s-1 trains
                if pending is not None:
                    crops = list(pending)
                    aug = self.aug(crops) if self.aug is not None else crops
                    images, aug_images, masks = self.transform(crops, aug)
                    yield ([images, aug_images, masks], torch.ones(1), torch.ones(1))
                pending = nxt


def main(args):
    import dig_amd.utils as utils
    from dig_amd.datasets import GpuBatchTransform, RandomMaskingGenerator
    from dig_amd.engine_for_pretraining_moco import train_one_epoch
    from dig_amd.optim_factory import create_optimizer
    from dig_amd.parallel import DistributedDataParallel
    from dig_amd.registry import create_model

    utils.init_distributed_mode(args)
    print(args)
    device = torch.device(args.device, getattr(args, "gpu", 0)) if args.device == "cuda" else torch.device(args.device)
    seed = args.seed + utils.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)

    model = create_model(args.model, pretrained=False, drop_path_rate=args.drop_path, drop_block_rate=None, mlp_dim=args.moco_mlp_dim,
                         dim=args.moco_dim, T=args.moco_t, num_windows=args.num_windows, encoder_type=args.encoder_type,
                         queue_size=args.queue_size, patchnet_name=args.patchnet_name)
    patch_size = model.encoder.patch_embed.patch_size
    print("Patch size = %s" % str(patch_size))
    args.window_size = (args.input_h // patch_size[0], args.input_w // patch_size[1])
    args.patch_size = patch_size
    model.to(device)
    n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print("number of params: {} M".format(n_parameters / 1e6))

    world, rank = utils.get_world_size(), utils.get_rank()
    data_path = [args.data_path] if isinstance(args.data_path, str) else list(args.data_path)
    data_path = [d for d in data_path if d]
    if data_path == ["synthetic"] and args.synthetic <= 0:
        args.synthetic = 100 * args.batch_size * world
    elif data_path and data_path != ["synthetic"]:
        for d in data_path:
            if os.path.isfile(os.path.join(d, "data.mdb")):
                raise SystemExit(f"--data_path {d}: an LMDB environment.  The LMDB reader (dataset/dataset_image.py: lmdb + cv2 + imgaug) stays with "
                                 "the reference; feed its uint8 crops to dig_amd.datasets.GpuBatchTransform (INTEGRATION.md, 'Input transform'), or "
                                 "point --data_path / --image_alone_path at a directory tree of image files")
        # (the reference mixes --data_path and --image_alone_path datasets when both are given: one list of directory trees here)
        alone = args.image_alone_path if isinstance(args.image_alone_path, (list, tuple)) else ([args.image_alone_path] if args.image_alone_path else [])
        args.image_alone_path = data_path + list(alone)
    if args.synthetic > 0:
        gen = RandomMaskingGenerator(args.window_size, args.mask_ratio, num_view=args.num_view, seed=seed, device=device)
        loader = SyntheticCrops(args.synthetic // world, args.batch_size, device, seed, gen)
    else:
        aug = None
        if args.aug_module:
            mod, fn = args.aug_module.rsplit(".", 1)
            aug = getattr(importlib.import_module(mod), fn)
        roots = args.image_alone_path if isinstance(args.image_alone_path, (list, tuple)) else [args.image_alone_path]
        loader = ImageFolderCrops(roots, args.batch_size, rank, world, GpuBatchTransform(args, seed=seed, device=device), aug, args.num_workers,
                                  cap=min(args.num_samples, args.aloneimage_num_samples))
    steps_per_epoch = len(loader)
    if steps_per_epoch == 0:
        raise SystemExit("fewer samples than one global batch")

    total_batch_size = args.batch_size * world
    args.lr = args.lr * total_batch_size / 256
    print("LR = %.8f" % args.lr)
    print("Batch size = %d" % total_batch_size)
    print("Number of training steps = %d" % steps_per_epoch)

    model_without_ddp = model
    if args.distributed:
        model = DistributedDataParallel(model)                 # SyncBN statistics, bucketed gradient all-reduce, key all-gather: dig_amd/parallel.py
    optimizer = create_optimizer(args, model_without_ddp)
    loss_scaler = utils.NativeScalerWithGradNormCount()
    lr_schedule_values = utils.cosine_scheduler(args.lr, args.min_lr, args.epochs, steps_per_epoch, warmup_epochs=args.warmup_epochs,
                                                warmup_steps=args.warmup_steps)
    if args.weight_decay_end is None:
        args.weight_decay_end = args.weight_decay
    wd_schedule_values = utils.cosine_scheduler(args.weight_decay, args.weight_decay_end, args.epochs, steps_per_epoch)
    print("Max WD = %.7f, Min WD = %.7f" % (max(wd_schedule_values), min(wd_schedule_values)))
    utils.auto_load_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler)

    # rank 0 writes the step scalars (loss, lr, weight decay, gradient norm) when --log_dir is given  (run_mae_pretraining_moco.py:357-361)
    log_writer = None
    if utils.get_rank() == 0 and args.log_dir is not None:
        os.makedirs(args.log_dir, exist_ok=True)
        log_writer = utils.TensorboardLogger(log_dir=args.log_dir)

    print(f"Start training for {args.epochs} epochs")
    start_time = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        if hasattr(loader, "set_epoch"):
            loader.set_epoch(epoch)
        if log_writer is not None:
            log_writer.set_step(epoch * steps_per_epoch)
        train_stats = train_one_epoch(model, None, None, loader, None, optimizer, device, epoch, loss_scaler, args.clip_grad, log_writer=log_writer,
                                      start_steps=epoch * steps_per_epoch, lr_schedule_values=lr_schedule_values,
                                      wd_schedule_values=wd_schedule_values, momentum_schedule=None, patch_size=patch_size[0],
                                      normlize_target=args.normlize_target, args=args)
        if args.output_dir and ((epoch + 1) % args.save_ckpt_freq == 0 or epoch + 1 == args.epochs):
            utils.save_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler, epoch=epoch)
        log_stats = {**{f"train_{k}": v for k, v in train_stats.items()}, "epoch": epoch, "n_parameters": n_parameters}
        if args.output_dir and utils.is_main_process():
            if log_writer is not None:
                log_writer.flush()
            with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps(log_stats) + "\n")
    print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))


if __name__ == "__main__":
    opts = get_args()
    if opts.output_dir:
        os.makedirs(opts.output_dir, exist_ok=True)
    main(opts)
