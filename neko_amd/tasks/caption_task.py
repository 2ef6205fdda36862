"""Caption / VQA task data side: ``gato/tasks/caption_task.py`` and
``gato/tasks/vqa_task.py`` over pre-tokenised in-memory datasets.

The reference reads img2dataset tar shards / COCO-VQA json + jpg folders (webdataset, PIL) and tokenises with the
``gpt2`` tokenizer; here an item already carries the image tensor uint8 ``(1, 3, H, W)`` (what ``process_data`` builds,
caption_task.py:100-104) and token ids.  Kept literally: the Python ``random.randint`` call sequence of ``sample_batch``
(caption_task.py:112-118, vqa_task.py:85-98: example index, then -- VQA -- the answer index) and of ``evaluate``
(:120-159 / :100-141), the batch-dict keys and the loss / perplexity arithmetic.  Pinned against the imported reference
(constructor bypassed, real HF fast tokenizer) by fixture G11 (tests/golden/make_fixture_caption_vqa.py).
"""
from __future__ import annotations

import random
from typing import Dict, List, Sequence

import torch


def _eval_metrics(total_loss: float, n: int) -> dict:
    avg = total_loss / n
    return {"loss": avg, "perplexity": torch.exp(torch.tensor(avg)).item()}


class TokenCaptionTask:
    """items: {'image': uint8 (1,3,H,W), 'text': [token ids]}"""
    kind = "caption"

    def __init__(self, dataset: Dict[str, Sequence[dict]], name: str = "caption"):
        assert "train" in dataset
        self.dataset, self.name = dataset, name

    def sample_batch(self, batch_size: int) -> List[dict]:
        idx = [random.randint(0, len(self.dataset["train"]) - 1) for _ in range(batch_size)]
        return [{"images": self.dataset["train"][i]["image"], "text": list(self.dataset["train"][i]["text"])} for i in idx]

    def evaluate(self, model, num_examples_to_test: int = 50, deterministic: bool = True,
                 log_examples_to_output: bool = False) -> dict:
        pol = getattr(model, "module", model)
        test = self.dataset["test"]
        num_examples_to_test = min(num_examples_to_test, len(test))
        picked = [test[random.randint(0, len(test) - 1)] for _ in range(num_examples_to_test)]
        total = 0.0
        for ex in picked:
            target = list(ex["text"])
            logits, _ = pol.predict_caption(ex["image"], max_length=len(target), deterministic=deterministic)
            tgt = torch.tensor(target, dtype=torch.long, device=logits.device)
            total += torch.nn.functional.cross_entropy(logits.float(), tgt).item()
        return _eval_metrics(total, num_examples_to_test)


class TokenVqaTask:
    """items: {'image': uint8 (1,3,H,W), 'question': [ids], 'answers': [[ids], ...]}; the training text is question +
    one randomly chosen answer (the reference tokenises the two joined by a space, vqa_task.py:93-96)."""
    kind = "vqa"

    def __init__(self, dataset: Dict[str, Sequence[dict]], name: str = "vqa"):
        assert "train" in dataset
        self.dataset, self.name = dataset, name

    def sample_batch(self, batch_size: int) -> List[dict]:
        idx = [random.randint(0, len(self.dataset["train"]) - 1) for _ in range(batch_size)]
        out = []
        for i in idx:
            item = self.dataset["train"][i]
            a = random.randint(0, len(item["answers"]) - 1)
            out.append({"images": item["image"], "text": list(item["question"]) + list(item["answers"][a])})
        return out

    def evaluate(self, model, num_examples_to_test: int = 50, deterministic: bool = True,
                 log_examples_to_output: bool = False) -> dict:
        pol = getattr(model, "module", model)
        test = self.dataset["test"]
        num_examples_to_test = min(num_examples_to_test, len(test))
        picked = [test[random.randint(0, len(test) - 1)] for _ in range(num_examples_to_test)]
        total = 0.0
        for ex in picked:
            a = random.randint(0, len(ex["answers"]) - 1)
            target = list(ex["answers"][a])
            # predict_answer(image, question_string) == predict_response(image, prompt_tokens=encode(question))
            logits, _ = pol.predict_response(ex["image"], prompt_tokens=list(ex["question"]), max_length=len(target),
                                             deterministic=deterministic)
            tgt = torch.tensor(target, dtype=torch.long, device=logits.device)
            total += torch.nn.functional.cross_entropy(logits.float(), tgt).item()
        return _eval_metrics(total, num_examples_to_test)
