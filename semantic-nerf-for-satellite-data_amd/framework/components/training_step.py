"""mirror of framework/components/training_step.py:5-11"""
import abc


class BaseTrainingStep:
    @abc.abstractmethod
    def training_step(self, pipeline, batch, batch_idx):
        pass

    def after_training_step(self, pipeline, outputs, batch, batch_idx):
        pass
