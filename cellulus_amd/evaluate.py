"""Evaluation against ground truth — same outputs as ``cellulus/evaluate.py:9-105``
(per-sample F1 / SEG, ``results_bandwidth-<b>.txt``).  The reference builds the
IoU table with O(#pred x #gt) full-image mask passes; one joint histogram of
(prediction, ground-truth) id pairs gives the identical table."""

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


def compute_pairwise_IoU(prediction, groundtruth):
    prediction_ids, p_inv = np.unique(prediction, return_inverse=True)
    groundtruth_ids, g_inv = np.unique(groundtruth, return_inverse=True)
    n_gt = int((groundtruth_ids != 0).sum())
    if n_gt == 0:
        return None
    joint = np.zeros((len(prediction_ids), len(groundtruth_ids)), dtype=np.int64)
    np.add.at(joint, (np.asarray(p_inv).ravel(), np.asarray(g_inv).ravel()), 1)
    p_size = joint.sum(axis=1, keepdims=True)
    g_size = joint.sum(axis=0, keepdims=True)
    p_keep, g_keep = prediction_ids != 0, groundtruth_ids != 0
    inter = joint[p_keep][:, g_keep].astype(float)
    union = (p_size[p_keep] + g_size[:, g_keep]) - inter
    IoU_table = inter / union
    IoG_table = inter / g_size[:, g_keep]
    return IoU_table, np.sum(IoU_table[IoG_table > 0.5]), n_gt


def compute_F1(IoU_table, threshold=0.5):
    IoU_table_thresholded = IoU_table > threshold
    FP = np.sum(np.sum(IoU_table_thresholded, axis=1) == 0)
    FN = np.sum(np.sum(IoU_table_thresholded, axis=0) == 0)
    TP = IoU_table.shape[1] - FN
    return 2 * TP / (2 * TP + FP + FN), TP, FP, FN
