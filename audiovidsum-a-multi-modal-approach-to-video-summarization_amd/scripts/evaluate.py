"""Counterpart of the reference's scripts/evaluate.py:6-42: score every video of a dataset with the model (one
video per call, B = 1, on the MI355X), then the per-video mean-threshold F1 and rank correlations, averaged.
Same signature and return keys as the reference; the metric arithmetic lives in evaluation.metrics."""
import torch

from ..evaluation.metrics import summarize_scores


def predict_dataset(model, dataset):
    """[(pred [S], target [S])] as numpy arrays, model in eval mode, no autograd."""
    model.eval()
    pairs = []
    with torch.no_grad():
        for features, scores in dataset:
            inputs = [features[k].unsqueeze(0).cuda() for k in ("visual", "audio")]
            pairs.append((model(*inputs).cpu().squeeze().numpy(), scores.numpy()))
    return pairs


def evaluate(model, dataset):
    return summarize_scores(predict_dataset(model, dataset))
