"""Evaluation path: forward-only inference and the reference's mIoU definition.

  test_step      utils/pipelines/trainer_lighting_bev.py:265-323  -- per-scan per-class Jaccard over ALL points
                 (sklearn.metrics.jaccard_score with labels 0..C-1, so points labelled -1 still enlarge the union
                 of the class they are predicted as), -1 for classes absent from the scan's labels
  test_epoch_end utils/pipelines/trainer_lighting_bev.py:325-383  -- -1 -> NaN, nan-mean over scans per class,
                 x100, nan-mean over classes
Everything stays on the device (the reference moves predictions to the CPU for sklearn)."""
import torch

from . import me as ME


def per_class_iou(preds, labels, num_classes=7, ignore_label=-1):
    """[C] IoU per class, -1 where the class does not occur in `labels`."""
    preds, labels = preds.long().view(-1), labels.long().view(-1)
    iou = torch.empty(num_classes, dtype=torch.float64, device=preds.device)
    cls = torch.arange(num_classes, device=preds.device).view(-1, 1)
    p, l = preds.view(1, -1) == cls, labels.view(1, -1) == cls
    inter = (p & l).sum(dim=1).double()
    union = (p | l).sum(dim=1).double()
    iou = torch.where(union > 0, inter / union.clamp(min=1), torch.zeros_like(inter))   # zero_division=0
    present = l.any(dim=1)
    return torch.where(present, iou, -torch.ones_like(iou))


def mean_iou(per_scan_iou):
    """[n_scans, C] with -1 for absent classes -> (per-class IoU in percent, mean IoU), NaN-aware"""
    x = per_scan_iou.clone().double()
    x[x == -1] = float("nan")
    per_class = torch.nanmean(x, dim=0) * 100
    return per_class, torch.nanmean(per_class)


@torch.no_grad()
def predict(model, coords, feats, manager=None):
    """validation / test forward: is_train=False (no BEV head, running BN statistics), arg-max class per voxel.
    After the first call the coordinate maps of a batch are built in one go from the recorded trace of map uses
    (ME.CoordinateManager.prepare: one host synchronisation instead of one per kernel map); `manager`: a manager
    prepared ahead of time for these coordinates (Predictor / evaluate() build the next batch's while the current
    forward pass runs)."""
    was_training = model.training
    model.eval()
    trace = getattr(model, "_lidog_eval_trace", None)
    if manager is None and trace is not None:
        manager = ME.CoordinateManager.prepare(coords, trace)
    if manager is not None:
        st = ME.SparseTensor(features=feats, coordinates=coords, coordinate_manager=manager)
    else:
        st = ME.SparseTensor(coordinates=coords, features=feats)
    out = model(st)
    model._lidog_eval_trace = st.coordinate_manager.trace
    logits = (out[0] if isinstance(out, tuple) else out).F
    model.train(was_training)
    return logits.max(dim=1)[1], logits


class Predictor:
    """predict() over a stream of batches with the NEXT batch's coordinate maps built on the side stream while
    the current forward pass runs: p = Predictor(model); preds, logits = p(coords, feats, next_coords)"""

    def __init__(self, model):
        self.model, self._next = model, None

    def __call__(self, coords, feats, next_coords=None):
        mgr = None
        if self._next is not None and self._next[0] is coords:
            mgr = self._next[1]
        self._next = None
        # the next batch's coordinates are valid NOW: its maps only wait for what is queued so far, not for the
        # forward pass that is about to be launched
        ready = torch.cuda.Event()
        ready.record()
        out = predict(self.model, coords, feats, mgr)
        trace = getattr(self.model, "_lidog_eval_trace", None)
        if next_coords is not None and trace is not None:
            self._next = (next_coords, ME.CoordinateManager.prepare(next_coords, trace, ready))
        return out


@torch.no_grad()
def evaluate(model, batches, num_classes=7, ignore_label=-1):
    """batches: iterable of dicts with coords_int [N,4], source_features0, source_sem_labels0 (one IoU row per
    scan, as test_step is called with batch size 1 in eval_target.py)"""
    rows = []
    run = Predictor(model)
    batches = list(batches)
    for i, b in enumerate(batches):
        coords = b["coords_int"]
        nxt = batches[i + 1]["coords_int"] if i + 1 < len(batches) else None
        preds, _ = run(coords, b["source_features0"], nxt)
        labels = b["source_sem_labels0"]
        for s in range(int(coords[:, 0].max().item()) + 1):
            sel = coords[:, 0] == s
            rows.append(per_class_iou(preds[sel], labels[sel], num_classes, ignore_label))
    return mean_iou(torch.stack(rows))


def write_results_csv(save_dir, source_names, target_name, per_scan_iou, class_names, first_target=True):
    """The result file of test_epoch_end (utils/pipelines/trainer_lighting_bev.py:325-383):
    `<save_dir>/results/<source>-TO-<target>.csv`, appended; header `source,target,<class names>,mean` before the first
    target's row; per-class IoU = nan-mean over scans (-1 = class absent from the scan) x 100, rounded to 2 decimals with a
    decimal COMMA, last column the nan-mean over classes.  `per_scan_iou`: [n_scans, C] as returned by per_class_iou;
    `class_names`: the C names (the reference takes `training_dataset.class2names[1:]`).  Returns the path."""
    import csv
    import os
    import numpy as np
    os.makedirs(os.path.join(save_dir, "results"), exist_ok=True)
    path = os.path.join(save_dir, "results", f"{source_names}-TO-{target_name}.csv")
    x = per_scan_iou.detach().double().cpu().numpy().copy()
    x[x == -1] = np.nan
    per_class = np.nanmean(x, axis=0) * 100
    average = np.nanmean(per_class, axis=0)
    with open(path, "a") as f:
        w = csv.writer(f)
        if first_target:
            w.writerow(["source", "target"] + list(class_names) + ["mean"])
        w.writerow([source_names, target_name] + [str(round(p, 2)).replace(".", ",") for p in per_class] +
                   [str(round(float(average), 2)).replace(".", ",")])
    return path
