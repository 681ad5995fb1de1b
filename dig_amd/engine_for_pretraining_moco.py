"""train_one_epoch for the MI355X engine -- same signature, argument meaning, meter names and error behaviour as the
reference step engine (engine_for_pretraining_moco.py:26-204).

Per step: per-iteration lr / weight-decay (:60-66) and MoCo momentum (:69-73) from the host schedules, MIM target
build (:83-111) as one gather kernel with bit-exact index order, model forward (one fused autograd node), loss
combine (:119-144), backward + global grad norm + fused AdamW through the loss_scaler object (:152-157), metrics."""
import math
import sys
import time
from typing import Iterable

import numpy as np
import torch

from . import ops
from . import step_graph
from . import utils


class _MimMSE(torch.autograd.Function):
    """F.mse_loss(vis_out, target, 'mean') with the masked-patch target gathered on the fly
    (un-normalise + '(p1 p2 c)' patchify + boolean select, engine_for_pretraining_moco.py:85-111,141)."""

    @staticmethod
    def forward(ctx, vis_out, images, idx, normalize=False):
        B, per, C = vis_out.shape
        M = B * per
        target = ops.mim_target(images, idx, M, 8, 32, normalize)
        loss = torch.zeros(1, device=vis_out.device, dtype=torch.float32)
        dpred = torch.empty((M, C), device=vis_out.device, dtype=torch.bfloat16)
        if vis_out.stride(2) != 1 or vis_out.stride(0) != per * vis_out.stride(1):
            vis_out = vis_out.contiguous()
        ops.mse_fwd_bwd(vis_out, vis_out.stride(1), target, M, C, 1.0, loss, dpred, C)
        ctx.dpred = dpred
        ctx.shape = (B, per, C)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        ops.cast_bf16_to_f32(ctx.dpred, d)
        ops.scale_by_device_scalar(d, g.reshape(1).float())
        return d, None, None, None


def mim_mse_loss(vis_out, images, idx, normalize=False):
    return _MimMSE.apply(vis_out, images, idx, normalize)


class _StepReadback:
    """Pinned-host mirror of the per-step device scalars, consumed one step behind the launch front."""
    FIELDS = 10

    def __init__(self, metric_logger, log_writer, core):
        self.metric_logger, self.log_writer, self.core = metric_logger, log_writer, core
        self.pending = []
        self.ring = [torch.empty(self.FIELDS, dtype=torch.float32).pin_memory() for _ in range(3)]
        self.n = 0

    def push(self, dev_vals, host_vals):
        buf = self.ring[self.n % len(self.ring)]
        self.n += 1
        buf.copy_(dev_vals, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, buf, host_vals))

    def resolve(self, keep):
        while len(self.pending) > keep:
            ev, buf, hv = self.pending.pop(0)
            ev.synchronize()
            host = buf.tolist()
            loss_value = host[0]
            if hv["w_contrast"] == 0.0 and not math.isfinite(host[1]):
                loss_value = float('nan')
            if self.core.use_pixel_target and (host[7] != host[8] or int(host[7]) != self.core._per_sample_mask):
                raise RuntimeError("masks must select the same number of tokens in every sample "
                                   f"(got {int(host[7])}..{int(host[8])}, expected {self.core._per_sample_mask}; after a deliberate change "
                                   "of the mask ratio call model.reset_mask_count())")
            if not math.isfinite(loss_value):
                print("Loss is {}, stopping training".format(loss_value))
                sys.exit(1)
            grad_norm = host[9] if hv["has_grad_norm"] else None
            ml = self.metric_logger
            # (engine_for_pretraining_moco.py:119-144: each meter only `if '<key>' in out_dict` -- a single-objective model logs its own loss)
            if self.core.use_moco_target:
                ml.update(loss_contrast=host[1], q1_acc1=host[3], q1_acc5=host[4], q2_acc1=host[5], q2_acc5=host[6])
            if self.core.use_pixel_target:
                ml.update(loss_pixel=host[2])
            ml.update(loss=loss_value)
            ml.update(loss_scale=hv["loss_scale"])
            ml.update(lr=hv["lr"])
            ml.update(min_lr=hv["min_lr"])
            ml.update(weight_decay=hv["weight_decay"])
            ml.update(grad_norm=grad_norm)
            lw = self.log_writer
            if lw is not None:
                lw.update(loss=loss_value, head="loss")
                lw.update(loss_scale=hv["loss_scale"], head="opt")
                lw.update(lr=hv["lr"], head="opt")
                lw.update(min_lr=hv["min_lr"], head="opt")
                lw.update(weight_decay=hv["weight_decay"], head="opt")
                lw.update(grad_norm=grad_norm, head="opt")
                lw.set_step()


def _step_body(model, core, optimizer, loss_scaler, max_norm, args, normlize_target, images, aug_images, bool_vis_masked_pos, moco_m,
               w_contrast, contrast_on, adamw_dev_scalars=None):
    """One iteration's device work (engine_for_pretraining_moco.py:76-157): forward, both losses, zero_grad, backward, grad norm,
    AdamW.  Nothing here reads back, so the same code runs eagerly (moco_m / w_contrast Python floats) or under HIP-graph capture
    (moco_m = device [m, 1-m], w_contrast = device scalar, AdamW scalars from device memory).  Returns the ten logged values as one
    device vector and whether a gradient norm is among them."""
    B = images.shape[0]
    prepared = None
    if bool_vis_masked_pos.is_cuda and args.num_view == 2 and bool_vis_masked_pos.is_contiguous():
        # the bool cast, the view-1 fill (:103-104) and the model's view-major uint8 copy in ONE launch (dig_mask_views_u8); the model takes
        # the prepared [2 B, N] uint8 rows as they are
        from . import ops
        prepared = ops.mask_views_u8(bool_vis_masked_pos.view(B, args.num_view, -1), 1 if args.only_mim_on_ori_img else args.num_view)
    if prepared is not None:
        bool_vis_masked_pos = prepared
    else:
        bool_vis_masked_pos = bool_vis_masked_pos.flatten(1).to(torch.bool).view(B, args.num_view, -1)
        if args.only_mim_on_ori_img:
            bool_vis_masked_pos[:, 1, :].fill_(0)                       # only the original view is masked (:103-104)

    out_dict = model(images, aug_images, bool_vis_masked_pos, moco_m, args.only_mim_on_ori_img)
    loss = 0.
    contra_loss = loss_pixel = None
    if 'contra_loss' in out_dict:               # (:119-133: absent for a Gen-only model)
        contra_loss = out_dict['contra_loss']
        if contrast_on:
            loss = loss + contra_loss * w_contrast
        # else: the reference adds 0 * contra_loss (:137), which leaves the value unchanged and back-propagates exact zeros
        # through the whole contrastive branch; leaving the term out lets the engine skip those launches (engine_core.backward).
        # A non-finite contra_loss would have poisoned the reference's loss (0 * inf = nan): the readback keeps that exit.
    if 'vis_out' in out_dict:                   # (:135-144: absent for a Dis-only model)
        vis_out = out_dict['vis_out']
        if args.only_mim_on_ori_img:
            loss_pixel = mim_mse_loss(vis_out[0], core._last_images, core._last_idx, bool(normlize_target))
        else:
            # both views carry a masked-pixel loss (:138-141); every view's target comes from the ORIGINAL crops (:106-108)
            loss_pixel = 0.
            for i in range(args.num_view):
                loss_pixel = loss_pixel + (1. / args.num_view) * mim_mse_loss(vis_out[i], core._last_images, core._last_idx_views[i],
                                                                             bool(normlize_target))
        loss = loss + loss_pixel * args.loss_weight_pixel
    if not isinstance(loss, torch.Tensor):
        # (a Dis-only model before contrast_start_epoch: the reference's loss is 0 * contra_loss -- every gradient an exact zero)
        loss = contra_loss * 0.0

    optimizer.zero_grad()
    if adamw_dev_scalars is None:
        grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), create_graph=False)
    else:
        grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=None, create_graph=False, adamw_dev_scalars=adamw_dev_scalars)

    # Everything the reference reads back with .item() (loss :146, accuracies :130-135, grad_norm :183) goes to the
    # host as ONE asynchronous copy, resolved while the next step is already queued: the reference's blocking reads
    # and its per-step torch.cuda.synchronize() (:159) drain the HIP queues twice per step and leave the GPU waiting
    # on kernel launches (measured: 1.1 ms of a 28.7 ms step).  The non-finite-loss exit (:148-150) and the
    # ragged-mask check therefore fire one step late -- before anything is logged or saved for that step.
    if contra_loss is None or loss_pixel is None:
        # a single-objective model: the absent meters travel as NaN and are not logged (_StepReadback.resolve)
        nan = torch.full((), float('nan'), device=loss.device)
        accs = [out_dict[k][0] for k in ('q1_acc1', 'q1_acc5', 'q2_acc1', 'q2_acc5')] if contra_loss is not None else [nan] * 4
        cnt = core._last_mask_counts if loss_pixel is not None else None
        dev_vals = torch.stack([loss.detach().reshape(()).float(), nan if contra_loss is None else contra_loss.detach().reshape(()),
                                nan if loss_pixel is None else loss_pixel.detach().reshape(())] + accs +
                               [nan if cnt is None else cnt.min().float(), nan if cnt is None else cnt.max().float(),
                                grad_norm.detach().reshape(()).float() if isinstance(grad_norm, torch.Tensor)
                                else torch.full((), float('nan') if grad_norm is None else float(grad_norm), device=loss.device)])
        return dev_vals, grad_norm is not None
    a1 = out_dict['q1_acc1']
    f32s = (loss, contra_loss, loss_pixel, a1)
    if (all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 for t in f32s) and loss.numel() == 1 and loss_pixel.numel() == 1
            and all(out_dict[k].data_ptr() == a1.data_ptr() + 4 * j for j, k in enumerate(('q1_acc1', 'q1_acc5', 'q2_acc1', 'q2_acc5')))
            and core._last_mask_counts.dtype == torch.int32 and core._last_mask_counts.is_contiguous()
            and (grad_norm is None or (isinstance(grad_norm, torch.Tensor) and grad_norm.dtype == torch.float32 and grad_norm.numel() == 1))):
        # one launch for the ten logged values (the accuracies are four consecutive floats of one tensor)
        from . import ops
        return ops.step_meters(loss.detach(), contra_loss.detach(), loss_pixel.detach(), a1, core._last_mask_counts,
                               None if grad_norm is None else grad_norm.detach()), grad_norm is not None
    dev_vals = torch.stack([loss.detach().reshape(()), contra_loss.detach().reshape(()), loss_pixel.detach().reshape(()),
                            out_dict['q1_acc1'][0], out_dict['q1_acc5'][0], out_dict['q2_acc1'][0], out_dict['q2_acc5'][0],
                            core._last_mask_counts.min().float(), core._last_mask_counts.max().float(),
                            grad_norm.detach().reshape(()).float() if isinstance(grad_norm, torch.Tensor)
                            else torch.full((), float('nan') if grad_norm is None else float(grad_norm), device=loss.device)])
    return dev_vals, grad_norm is not None


def train_one_epoch(model: torch.nn.Module, teacher_model, teacher_model_without_ddp, data_loader: Iterable,
                    word_data_loader: Iterable, optimizer: torch.optim.Optimizer, device: torch.device, epoch: int,
                    loss_scaler, max_norm: float = 0, patch_size: int = 16, normlize_target: bool = True, log_writer=None,
                    lr_scheduler=None, start_steps=None, lr_schedule_values=None, wd_schedule_values=None,
                    momentum_schedule=None, args=None):
    if args.num_view != 2:
        # the reference model itself is 2-view only: it stacks [image, aug_image] = 2B rows and chunks every result in two
        # (modeling_pretrain_moco_mim_ori.py:491,501,521,548), so a [B, num_view, N] mask with num_view != 2 fails in its encoder
        raise NotImplementedError("num_view must be 2: the model stacks exactly two views (modeling_pretrain_moco_mim_ori.py:491-523)")
    model.train()
    core = model.module if hasattr(model, "module") else model
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    metric_logger.add_meter('min_lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    print_freq = 100
    iters_per_epoch = len(data_loader)

    # contrast loss-weight warm-up (:48-56)
    if epoch == args.contrast_start_epoch:
        ws = min(args.contrast_warmup_steps, iters_per_epoch)
        contrast_loss_weights = np.linspace(0., args.loss_weight_contrast, ws)
        if ws < iters_per_epoch:
            contrast_loss_weights = np.hstack([contrast_loss_weights, np.ones(iters_per_epoch - ws) * args.loss_weight_contrast])
    elif epoch > args.contrast_start_epoch:
        contrast_loss_weights = np.ones(iters_per_epoch) * args.loss_weight_contrast
    else:
        contrast_loss_weights = np.zeros(iters_per_epoch)

    readback = _StepReadback(metric_logger, log_writer, core)
    graph_ok = step_graph.usable(core, model, optimizer, loss_scaler, max_norm)
    for step, (batch, text, text_lens) in enumerate(metric_logger.log_every(data_loader, print_freq, header)):
        it = start_steps + step
        if lr_schedule_values is not None or wd_schedule_values is not None:
            for param_group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    param_group["lr"] = lr_schedule_values[it] * param_group["lr_scale"]
                if wd_schedule_values is not None and param_group["weight_decay"] > 0:
                    param_group["weight_decay"] = wd_schedule_values[it]
        moco_m = utils.adjust_moco_momentum(epoch + 1.0 * step / iters_per_epoch, args) if args.use_moco_m_cos else args.moco_m
        metric_logger.update(moco_m=moco_m)

        t_host = time.perf_counter()
        images, aug_images, bool_vis_masked_pos = batch
        images = images.to(device, non_blocking=True)
        aug_images = aug_images.to(device, non_blocking=True)
        bool_vis_masked_pos = bool_vis_masked_pos.to(device, non_blocking=True)
        w_contrast = float(contrast_loss_weights[step])
        loss_scale_value = loss_scaler.state_dict()["scale"]
        if graph_ok and getattr(core, "_per_sample_mask", None) is not None:
            # the captured form of this launch sequence (dig_amd/step_graph.py): scalars and inputs go to static device buffers,
            # then one graph launch; the first steps of every sequence run the same body eagerly
            sg = step_graph.get(core, images.device)
            m32 = float(np.float32(moco_m))                            # dig_ema_update's own arithmetic: (float)(1 - (double)(float)m)
            sg.set_scalars(step_graph.adamw_scalars(optimizer) + [m32, 1.0 - m32, w_contrast])
            sig, (s_img, s_aug, s_mask) = sg.set_inputs(images, aug_images, bool_vis_masked_pos)
            key = (sig, w_contrast != 0.0, bool(normlize_target), core._per_sample_mask, float(args.loss_weight_pixel), bool(args.only_mim_on_ori_img),
                   bool(getattr(core, "overlap_streams", True)))
            dev_vals = sg.run(key, lambda: _step_body(model, core, optimizer, loss_scaler, max_norm, args, normlize_target, s_img, s_aug, s_mask,
                                                      sg.scalars[6:8], sg.scalars[8], w_contrast != 0.0, sg.scalars[0:6])[0])
            optimizer._step += 1
            has_grad_norm = True
        else:
            dev_vals, has_grad_norm = _step_body(model, core, optimizer, loss_scaler, max_norm, args, normlize_target, images, aug_images,
                                                 bool_vis_masked_pos, moco_m, w_contrast, w_contrast != 0.0)
        min_lr, max_lr = 10., 0.
        for group in optimizer.param_groups:
            min_lr, max_lr = min(min_lr, group["lr"]), max(max_lr, group["lr"])
        weight_decay_value = None
        for group in optimizer.param_groups:
            if group["weight_decay"] > 0:
                weight_decay_value = group["weight_decay"]
        readback.push(dev_vals, dict(loss_scale=loss_scale_value, lr=max_lr, min_lr=min_lr, weight_decay=weight_decay_value,
                                     has_grad_norm=has_grad_norm, w_contrast=w_contrast))
        # host seconds spent queueing this step (diagnostic: bench.py reports it as host_ms_per_step)
        core._host_launch = (getattr(core, "_host_launch", (0.0, 0))[0] + time.perf_counter() - t_host, getattr(core, "_host_launch", (0.0, 0))[1] + 1)
        readback.resolve(keep=1)
        if lr_scheduler is not None:
            lr_scheduler.step_update(start_steps + step)
        if step % print_freq == 0 or step == iters_per_epoch - 1:
            readback.resolve(keep=0)                                        # the logger prints after this iteration
        if step >= 1 and step % (args.eval_freq * 10) == 0:
            readback.resolve(keep=0)
            utils.save_model(args=args, model=model, model_without_ddp=core, optimizer=optimizer, loss_scaler=loss_scaler,
                             epoch="{0}_{1}".format(epoch, step))
        sys.stdout.flush()

    readback.resolve(keep=0)
    torch.cuda.synchronize()
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
