"""Text-task data side: ``gato/tasks/text_task.py`` over a pre-tokenised corpus.

The reference's ``TextTask`` loads HF datasets by name and the ``gpt2`` tokenizer (network); here the corpus is a
dict ``{'train': [[ids...], ...], 'test': [...]}`` of already tokenised documents.  What is kept literally:
``sample_batch`` (text_task.py:32-59: ``np.random.randint`` document draw WITH replacement, every document cut into
consecutive ``context_length`` chunks -- the tokenizer call's ``truncation=True, max_length, return_overflowing_tokens``
-- empty chunks skipped, the first ``batch_size`` chunks kept) and ``evaluate`` (:61-114: random split point,
``predict_text`` on the prefix, mean cross-entropy of the continuation, perplexity).  Pinned against the imported
reference driven by a real HF fast tokenizer and an in-memory ``datasets.Dataset`` (fixture G10,
tests/golden/make_fixture_text_task.py).
"""
from __future__ import annotations

import copy
from typing import Dict, List, Sequence

import numpy as np
import torch


class TokenTextTask:
    kind = "text"

    def __init__(self, corpus: Dict[str, Sequence[Sequence[int]]], context_length: int, name: str = "text"):
        assert "train" in corpus, "corpus needs a 'train' partition ('test' for evaluate)"
        self.text_dataset = {k: [list(map(int, d)) for d in v] for k, v in corpus.items()}
        self.context_length = int(context_length)
        self.name = name

    def _chunks(self, doc: List[int]) -> List[List[int]]:
        n = self.context_length
        return [doc[i:i + n] for i in range(0, len(doc), n)] or [[]]

    def sample_batch(self, batch_size: int, is_test: bool = False) -> List[dict]:
        """text_task.py:32-59."""
        docs = self.text_dataset["train" if not is_test else "test"]
        idx = np.random.randint(0, len(docs), size=batch_size)
        out: List[dict] = []
        for i in idx:
            for ids in self._chunks(docs[int(i)]):
                if len(ids) > 0:
                    out.append({"text": ids, "images": None, "continuous_obs": None, "discrete_obs": None,
                                "continuous_actions": None, "discrete_actions": None})
                    if len(out) == batch_size:
                        return out
        return out

    def evaluate(self, model, num_examples_to_test: int = 50, deterministic: bool = True,
                 log_examples_to_output: bool = False) -> dict:
        """text_task.py:61-114."""
        pol = getattr(model, "module", model)
        num_examples_to_test = min(num_examples_to_test, len(self.text_dataset["test"]))
        batch_dicts = self.sample_batch(num_examples_to_test, is_test=True)
        total_loss, tested = 0.0, 0
        for idx in range(min(num_examples_to_test, len(batch_dicts))):
            tokens = batch_dicts[idx]["text"]
            ith = np.random.randint(1, len(tokens))
            new = copy.deepcopy(batch_dicts[idx])
            new["text"] = tokens[:ith]
            target = tokens[ith:]
            pred_logits, pred_tokens = pol.predict_text(new, max_length=len(target), deterministic=deterministic)
            if log_examples_to_output and idx % 50 == 0:
                tk = pol.text_tokenizer
                print(f"Text Example : {tk.decode(tokens)} \n Input passed to model : {tk.decode(new['text'])} \n "
                      f"Predicted output : {tk.decode(pred_tokens)}")
            tgt = torch.tensor(target, dtype=torch.long, device=pred_logits.device)
            total_loss += torch.nn.functional.cross_entropy(pred_logits.float(), tgt).item()
            tested += 1
        avg = total_loss / tested
        return {"loss": avg, "perplexity": torch.exp(torch.tensor(avg)).item()}
