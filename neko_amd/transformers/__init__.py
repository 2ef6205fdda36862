from .trajectory_gpt2 import GPT2Model, GPT2Config  # reference: gato/transformers/__init__.py re-exports GPT2Model
