"""Evaluation statistics of the reference (/root/reference/src/utilities/stats.py:7-61): per-class AP / AUC /
sampled PR and ROC curves plus the (class-independent) top-1 accuracy, and d' from the mean AUC."""
import numpy as np
from scipy import stats as _sstats
from sklearn import metrics


def d_prime(auc):
    return _sstats.norm().ppf(auc) * np.sqrt(2.0)


def calculate_stats(output, target, skip_auc=False, every=1000):
    output, target = np.asarray(output), np.asarray(target)
    acc = metrics.accuracy_score(np.argmax(target, 1), np.argmax(output, 1))
    out = []
    for k in range(target.shape[-1]):
        t, o = target[:, k], output[:, k]
        prec, rec, _ = metrics.precision_recall_curve(t, o)
        fpr, tpr, _ = metrics.roc_curve(t, o)
        out.append({"precisions": prec[::every], "recalls": rec[::every],
                    "AP": metrics.average_precision_score(t, o, average=None),
                    "fpr": fpr[::every], "fnr": 1.0 - tpr[::every],
                    "auc": None if skip_auc else metrics.roc_auc_score(t, o, average=None),
                    "acc": acc})
    return out


def summarize(stats_list, main_metric):
    """the 5 numbers the reference logs per epoch (traintest.py:195-218): main, mAUC, P, R, d'"""
    mAP = float(np.mean([s["AP"] for s in stats_list]))
    mAUC = float(np.mean([s["auc"] for s in stats_list]))
    acc = float(stats_list[0]["acc"])
    mid_p = float(np.mean([s["precisions"][len(s["precisions"]) // 2] for s in stats_list]))
    mid_r = float(np.mean([s["recalls"][len(s["recalls"]) // 2] for s in stats_list]))
    return {"mAP": mAP, "acc": acc, "mAUC": mAUC, "precision": mid_p, "recall": mid_r, "d_prime": float(d_prime(mAUC)),
            "main": mAP if main_metric == "mAP" else acc}
