"""`evaluate()` of the reference's fine-tune engine (engine_for_finetuning.py:214-285) on the MI355X: greedy decode through
`dig_amd.recognizer.RecModel`, SeqCrossEntropyLoss, string accuracy and character F-measure all on the device; one host read
per batch.  Same signature, meter names and return value.  `train_one_epoch` of that file (the fine-tune training step,
SURVEY.md 8f row N1) is not built."""
import torch

from . import utils
from .recognizer import SeqCrossEntropyLoss, accuracy, recognition_f_measure


def _vocabulary(dataset):
    """idx_to_class of the reference datasets (dataset/dataset_image.py:60-83) as a list indexed by class id."""
    if hasattr(dataset, "idx_to_class"):
        m = dataset.idx_to_class
        return [m[i] for i in range(len(m))]
    return list(dataset.voc)


@torch.no_grad()
def evaluate(data_loader, model, device, args=None):
    if getattr(args, "beam_width", 0):
        raise NotImplementedError("beam search is not built (greedy decode only)")
    criterion = SeqCrossEntropyLoss()
    metric_logger = utils.MetricLogger(delimiter="  ")
    header = 'Test:'
    model.eval()
    voc = _vocabulary(data_loader.dataset)
    for batch in metric_logger.log_every(data_loader, 10, header):
        images, target, lens = batch[0], batch[1], batch[-1]
        images = images.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        output, _, _, _ = model((images, target, lens))
        loss = criterion(output, target, lens)                                   # (on probabilities, as the reference does: :249)
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
