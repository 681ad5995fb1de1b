"""The reference's fine-tune engine (engine_for_finetuning.py) on the MI355X, same signatures, meter names and return values.

`evaluate()` (:214-285): greedy decode through `dig_amd.recognizer.RecModel`, SeqCrossEntropyLoss, string accuracy and character
F-measure all on the device; one host read per batch.
`train_one_epoch()` (:54-210) with `train_class_batch` (:26-46) for the torch.amp branch (`loss_scaler` given): per-step lr / weight
decay from the schedules times each group's `lr_scale`, teacher-forced forward, SeqCrossEntropyLoss, backward + grad norm + fused
AdamW through the scaler object, gradient accumulation (`update_freq`), class accuracy of the arg-max, evaluation every
`eval_freq` steps.  The meters travel in one pinned asynchronous copy resolved one step later (as in the pre-training engine).
Mixup, model EMA, the distillation teacher and the deepspeed branch are not built (they raise)."""
import math
import sys
from typing import Iterable, Optional

import torch

from . import utils
from .recognizer import SeqCrossEntropyLoss, accuracy, recognition_f_measure


def train_class_batch(model, samples, target, tgt_lens, criterion, criterion_aux=None, args=None, teacher_model=None, metric_logger=None):
    if teacher_model is not None or getattr(args, "use_seq_cls_token", False):
        raise NotImplementedError("feature distillation / seq-cls-token models are not built")
    outputs, _, _, _ = model((samples, target, tgt_lens))
    loss = criterion(outputs, target, tgt_lens)
    return loss, outputs, None


def train_one_epoch(model: torch.nn.Module, criterion: torch.nn.Module, data_loader: Iterable, optimizer, device: torch.device, epoch: int,
                    loss_scaler, max_norm: float = 0, model_ema=None, mixup_fn=None, log_writer=None, start_steps=None,
                    lr_schedule_values=None, wd_schedule_values=None, num_training_steps_per_epoch=None, update_freq=None, args=None,
                    data_loader_val=None, max_accuracy=0., criterion_aux=None, teacher_model=None):
    if loss_scaler is None:
        raise NotImplementedError("the deepspeed branch (loss_scaler=None) is not built")
    if mixup_fn is not None or teacher_model is not None or getattr(args, "w2v_path", None) is not None:
        raise NotImplementedError("mixup / distillation teacher / w2v targets are not built")
    update_freq = update_freq or 1
    model.train(True)
    core = model.module if hasattr(model, "module") else model
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    metric_logger.add_meter('min_lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    print_freq = 100
    voc = _vocabulary(data_loader.dataset) if hasattr(data_loader, "dataset") else None
    optimizer.zero_grad()
    ring = [torch.empty(3, dtype=torch.float32).pin_memory() for _ in range(3)]
    pending = []

    def resolve(keep):
        while len(pending) > keep:
            ev, buf, hv = pending.pop(0)
            ev.synchronize()
            loss_value, acc, gn = buf.tolist()
            if not math.isfinite(loss_value):
                print("Loss is {}, stopping training".format(loss_value))
                sys.exit(1)
            metric_logger.update(loss=loss_value)
            metric_logger.update(class_acc=acc if hv["has_acc"] else None)
            metric_logger.update(loss_scale=hv["loss_scale"])
            metric_logger.update(lr=hv["lr"])
            metric_logger.update(min_lr=hv["min_lr"])
            metric_logger.update(weight_decay=hv["weight_decay"])
            metric_logger.update(grad_norm=gn if hv["has_gn"] else None)
            if log_writer is not None:
                log_writer.update(loss=loss_value, head="loss")
                log_writer.update(class_acc=acc if hv["has_acc"] else None, head="loss")
                log_writer.update(loss_scale=hv["loss_scale"], head="opt")
                log_writer.update(lr=hv["lr"], head="opt")
                log_writer.update(min_lr=hv["min_lr"], head="opt")
                log_writer.update(weight_decay=hv["weight_decay"], head="opt")
                log_writer.update(grad_norm=gn if hv["has_gn"] else None, head="opt")
                log_writer.set_step()

    n_iter = len(data_loader)
    for data_iter_step, data in enumerate(metric_logger.log_every(data_loader, print_freq, header)):
        samples, targets, tgt_lens = data
        model.train(True)
        step = data_iter_step // update_freq
        if num_training_steps_per_epoch is not None and step >= num_training_steps_per_epoch:
            continue
        it = (start_steps or 0) + step
        if lr_schedule_values is not None or wd_schedule_values is not None and data_iter_step % update_freq == 0:
            for param_group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    param_group["lr"] = lr_schedule_values[it] * param_group["lr_scale"]
                if wd_schedule_values is not None and param_group["weight_decay"] > 0:
                    param_group["weight_decay"] = wd_schedule_values[it]
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        loss, output, _ = train_class_batch(model, samples, targets, tgt_lens, criterion, criterion_aux, args, teacher_model, metric_logger)
        loss_report = loss.detach()
        loss = loss / update_freq
        grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=None, create_graph=False,
                                update_grad=(data_iter_step + 1) % update_freq == 0)
        if (data_iter_step + 1) % update_freq == 0:
            optimizer.zero_grad()
            if model_ema is not None:
                model_ema.update(core)                                          # engine_for_finetuning.py:136-139
        loss_scale_value = loss_scaler.state_dict()["scale"]
        acc = accuracy(output.detach().argmax(-1), targets, voc) if voc is not None else None
        dev_vals = torch.stack([loss_report.reshape(()).float(), acc.float() if acc is not None else torch.zeros((), device=loss_report.device),
                                grad_norm.detach().reshape(()).float() if isinstance(grad_norm, torch.Tensor)
                                else torch.zeros((), device=loss_report.device)])
        min_lr, max_lr = 10., 0.
        weight_decay_value = None
        for group in optimizer.param_groups:
            min_lr, max_lr = min(min_lr, group["lr"]), max(max_lr, group["lr"])
            if group["weight_decay"] > 0:
                weight_decay_value = group["weight_decay"]
        buf = ring[data_iter_step % len(ring)]
        buf.copy_(dev_vals, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pending.append((ev, buf, dict(loss_scale=loss_scale_value, lr=max_lr, min_lr=min_lr, weight_decay=weight_decay_value,
                                      has_acc=acc is not None, has_gn=isinstance(grad_norm, torch.Tensor))))
        resolve(keep=1)
        if data_iter_step % print_freq == 0 or data_iter_step == n_iter - 1:
            resolve(keep=0)
        if step >= 1 and args is not None and step % args.eval_freq == 0 and data_loader_val is not None:
            resolve(keep=0)
            test_stats = evaluate(data_loader_val, core, device, args=args)
            print(f"Accuracy of the network on the test images: {test_stats['acc']:.4f}%")
            max_accuracy = max(max_accuracy, test_stats["acc"])
        sys.stdout.flush()
    resolve(keep=0)
    torch.cuda.synchronize()
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    train_stats = {k: meter.global_avg for k, meter in metric_logger.meters.items()}
    train_stats.update({'max_accuracy': max_accuracy})
    return train_stats


def _vocabulary(dataset):
    """idx_to_class of the reference datasets (dataset/dataset_image.py:60-83) as a list indexed by class id."""
    if hasattr(dataset, "idx_to_class"):
        m = dataset.idx_to_class
        return [m[i] for i in range(len(m))]
    return list(dataset.voc)


@torch.no_grad()
def evaluate(data_loader, model, device, args=None):
    beam = int(getattr(args, "beam_width", 0) or 0)                                  # engine_for_finetuning.py:246-256: with beam search the model
    criterion = SeqCrossEntropyLoss()                                               # returns token ids, the loss meter reads 0
    metric_logger = utils.MetricLogger(delimiter="  ")
    header = 'Test:'
    model.eval()
    voc = _vocabulary(data_loader.dataset)
    for batch in metric_logger.log_every(data_loader, 10, header):
        images, target, lens = batch[0], batch[1], batch[-1]
        images = images.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        output, _, _, _ = model((images, target, lens))
        if beam > 0:
            if getattr(model, "beam_width", 0) != beam:
                raise RuntimeError(f"args.beam_width={beam} but the model was built with beam_width={getattr(model, 'beam_width', 0)}")
            loss, pred_ids = torch.zeros((), device=output.device), output
        else:
            loss = criterion(output, target, lens)                               # (on probabilities, as the reference does: :249)
            pred_ids = output.argmax(-1)
        vals = torch.stack([loss.double(), accuracy(pred_ids, target, voc).double(), recognition_f_measure(pred_ids, target, voc)]).tolist()
        batch_size = images.shape[0]
        metric_logger.update(loss=vals[0])
        metric_logger.meters['acc'].update(vals[1], n=batch_size)
        metric_logger.meters['recognition_fmeasure'].update(vals[2], n=batch_size)
    metric_logger.synchronize_between_processes()
    print('* {n} images, Acc {acc.global_avg:.4f} loss {losses.global_avg:.4f} Rec_fmeasure {rec_f.global_avg:.4f}'
          .format(n=metric_logger.acc.count, acc=metric_logger.acc, losses=metric_logger.loss, rec_f=metric_logger.recognition_fmeasure))
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
