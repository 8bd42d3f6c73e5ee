from .deepspeech_trainer import DeepSpeechTrainer, Epochs, asr_metrics
