"""Evaluation against ground truth — same outputs as ``cellulus/evaluate.py:9-105``
(per-sample F1 / SEG, ``results_bandwidth-<b>.txt``).  The reference builds the
IoU table with O(#pred x #gt) full-image mask passes; one joint histogram of
(prediction, ground-truth) id pairs, taken on the device in a single pass over the two
label maps (``clx_joint_histogram``), gives the identical table."""

import numpy as np
from tqdm import tqdm

from .configs.inference_config import InferenceConfig
from .datasets.meta_data import DatasetMetaData
from .utils import zarr_io


def evaluate(inference_config: InferenceConfig) -> None:
    dataset_config = inference_config.dataset_config
    meta = DatasetMetaData.from_dataset_config(dataset_config)
    f = zarr_io.open(inference_config.evaluation_dataset_config.container_path)
    ds_segmentation = f[inference_config.evaluation_dataset_config.secondary_dataset_name]
    ds_groundtruth = f[inference_config.evaluation_dataset_config.dataset_name]
    for bandwidth in range(inference_config.num_bandwidths):
        sample_list, F1_list, SEG_list, TP_list, FP_list, FN_list = [], [], [], [], [], []
        SEG_dataset, n_ids_dataset = 0, 0
        for sample in tqdm(range(meta.num_samples)):
            groundtruth = ds_groundtruth[sample, 0].astype(np.uint16)
            prediction = ds_segmentation[sample, bandwidth].astype(np.uint16)
            returned_values = compute_pairwise_IoU(prediction, groundtruth)
            if returned_values is not None:
                IoU, SEG_image, n_GTids_image = returned_values
                F1_image, TP_image, FP_image, FN_image = compute_F1(IoU)
                F1_list.append(F1_image)
                SEG_list.append(SEG_image / n_GTids_image)
                SEG_dataset += SEG_image
                n_ids_dataset += n_GTids_image
                TP_list.append(TP_image)
                FP_list.append(FP_image)
                FN_list.append(FN_image)
                sample_list.append(sample)
                print(f"{sample}: F1={F1_image:.3f}, SEG={SEG_image/n_GTids_image:.3f}")
        F1_dataset = 2 * sum(TP_list) / (2 * sum(TP_list) + sum(FP_list) + sum(FN_list))
        print(f"F1 for dataset  is {F1_dataset:.05f}")
        print(f"SEG for dataset  is {SEG_dataset/n_ids_dataset:.05f}")
        with open(f"results_bandwidth-{bandwidth}.txt", "w") as out:
            out.writelines("file index, F1, SEG, TP, FP, FN \n")
            out.writelines("+++++++++++++++++++++++++++++++++\n")
            for i in range(len(sample_list)):
                out.writelines(f"{sample_list[i]}, {F1_list[i]:.05f}, {SEG_list[i]:.05f},"
                               f" {TP_list[i]}, {FP_list[i]}, {FN_list[i]}\n")
            out.writelines("+++++++++++++++++++++++++++++++++\n")
            out.writelines(f"F1 for complete dataset is {F1_dataset:.05f} \n")
            out.writelines(f"SEG for complete dataset is {SEG_dataset/n_ids_dataset:.05f} \n")


MAX_IDS = 1 << 16                      # label maps are stored as uint16 (detect.py:62-70, segment.py:24-32)


def joint_histogram_on_device(prediction, groundtruth, device=None):
    """(prediction ids, ground-truth ids, joint[#pred ids][#gt ids] int64) — ids ascending as
    ``np.unique`` returns them, background 0 included when present; one pass over the two maps
    on the device (``csrc/evaluate.hip``)."""
    import torch

    from . import _clx

    if torch.is_tensor(prediction):
        device = prediction.device
    elif device is None:
        if not torch.cuda.is_available():
            raise _clx.ClxError("evaluate needs a HIP device; cellulus_amd has no CPU path")
        device = torch.device("cuda", torch.cuda.current_device())

    def up(a):
        if not torch.is_tensor(a):
            a = torch.from_numpy(np.ascontiguousarray(a).astype(np.int32))
        return a.to(device=device, dtype=torch.int32).contiguous().reshape(-1)

    p, g = up(prediction), up(groundtruth)
    assert p.numel() == g.numel(), "prediction and ground truth differ in shape"
    _clx.require_device(p, "prediction")
    st = _clx.stream_ptr(device)
    present = torch.zeros(2, MAX_IDS, dtype=torch.int32, device=device)
    bad = torch.zeros(1, dtype=torch.int32, device=device)
    _clx.call("clx_label_presence", _clx.ptr(p), p.numel(), MAX_IDS, _clx.ptr(present[0]), _clx.ptr(bad), st)
    _clx.call("clx_label_presence", _clx.ptr(g), g.numel(), MAX_IDS, _clx.ptr(present[1]), _clx.ptr(bad), st)
    index = (torch.cumsum(present, dim=1, dtype=torch.int32) - 1).contiguous()   # id -> row / column
    count = present.sum(dim=1).tolist()
    if int(bad.item()):
        raise ValueError("label ids must lie in [0, 65536): the label maps are uint16 on disk")
    joint = torch.zeros(count[0], count[1], dtype=torch.int64, device=device)
    _clx.call("clx_joint_histogram", _clx.ptr(p), _clx.ptr(g), p.numel(), _clx.ptr(index[0]),
              _clx.ptr(index[1]), count[1], _clx.ptr(joint), st)
    ids = [torch.nonzero(present[k]).reshape(-1).cpu().numpy() for k in (0, 1)]
    return ids[0], ids[1], joint.cpu().numpy()


def iou_from_joint(prediction_ids, groundtruth_ids, joint):
    """IoU table, SEG sum and #gt ids of ``evaluate.py:72-100`` from the joint histogram:
    |P_j & G_k| = joint[j, k], |P_j| and |G_k| its row / column sums, |P_j | G_k| their sum minus
    the intersection; the divisions are the reference's (integer counts -> float64)."""
    n_gt = int((groundtruth_ids != 0).sum())
    if n_gt == 0:
        return None
    p_size = joint.sum(axis=1, keepdims=True)
    g_size = joint.sum(axis=0, keepdims=True)
    p_keep, g_keep = prediction_ids != 0, groundtruth_ids != 0
    inter = joint[p_keep][:, g_keep].astype(float)
    union = (p_size[p_keep] + g_size[:, g_keep]) - inter
    IoU_table = inter / union
    IoG_table = inter / g_size[:, g_keep]
    return IoU_table, np.sum(IoU_table[IoG_table > 0.5]), n_gt


def compute_pairwise_IoU(prediction, groundtruth, device=None):
    return iou_from_joint(*joint_histogram_on_device(prediction, groundtruth, device))


def compute_F1(IoU_table, threshold=0.5):
    IoU_table_thresholded = IoU_table > threshold
    FP = np.sum(np.sum(IoU_table_thresholded, axis=1) == 0)
    FN = np.sum(np.sum(IoU_table_thresholded, axis=0) == 0)
    TP = IoU_table.shape[1] - FN
    return 2 * TP / (2 * TP + FP + FN), TP, FP, FN
