from .dataset_config import DatasetConfig
from .experiment_config import ExperimentConfig
from .inference_config import InferenceConfig
from .model_config import ModelConfig
from .train_config import TrainConfig

__all__ = ["DatasetConfig", "ExperimentConfig", "InferenceConfig", "ModelConfig", "TrainConfig"]
