"""Mirror of the reference's scripts/evaluate.py:6-42: per-video mean-threshold F1 + Spearman + Kendall.
The model call runs on the MI355X (AVBiLSTMModel HIP path); the metrics stay on the host (numpy/scipy)."""
import numpy as np
import torch
from scipy.stats import kendalltau, spearmanr


def evaluate(model, dataset):
    model.eval()
    all_preds, all_targets = [], []
    with torch.no_grad():
        for features, scores in dataset:
            visual = features["visual"].unsqueeze(0).cuda()
            audio = features["audio"].unsqueeze(0).cuda()
            preds = model(visual, audio).cpu().squeeze()
            all_preds.append(preds.numpy())
            all_targets.append(scores.numpy())
    f1_scores, spearmans, kendalls = [], [], []
    for pred, target in zip(all_preds, all_targets):
        binary_pred = (pred > np.mean(pred)).astype(int)
        binary_target = (target > np.mean(target)).astype(int)
        tp = np.logical_and(binary_pred, binary_target).sum()
        precision = tp / binary_pred.sum()
        recall = tp / binary_target.sum()
        f1_scores.append(2 * (precision * recall) / (precision + recall + 1e-8))
        spearmans.append(spearmanr(pred, target).correlation)
        kendalls.append(kendalltau(pred, target).correlation)
    return {"f1": np.mean(f1_scores), "spearman": np.mean(spearmans), "kendall": np.mean(kendalls)}
