"""Generate tests/golden/g10_text_task.pt by running the REFERENCE's TextTask (gato/tasks/text_task.py).

Build container only.  The class is imported unmodified; its constructor (HF hub download of a dataset and of the gpt2
tokenizer) is bypassed with ``TextTask.__new__`` and the three attributes it would have set are supplied from local
objects: a REAL HF fast tokenizer (word-level, built in memory with the `tokenizers` library -- so truncation /
return_overflowing_tokens / return_length are HF's own semantics) and in-memory ``datasets.Dataset`` partitions.

    python tests/golden/make_fixture_text_task.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
VOCAB = 60


class FakePolicy:
    """predict_text with logits that are a fixed function of (prefix, step): pins evaluate()'s own arithmetic."""
    device = "cpu"

    def __init__(self):
        self.module = self
        self.text_tokenizer = None

    def predict_text(self, batch_dict, max_length=20, deterministic=True):
        prefix = batch_dict["text"]
        g = torch.Generator().manual_seed(1000 * len(prefix) + int(prefix[-1]) + max_length)
        logits = torch.randn(max_length, VOCAB, generator=g)
        return logits, list(torch.argmax(logits, dim=-1))


def main():
    import datasets
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    import ref_shims
    ref_shims.install(128)
    from gato.tasks.text_task import TextTask

    vocab = {f"w{i}": i for i in range(VOCAB)}
    vocab["[UNK]"] = VOCAB
    tok = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="[UNK]")

    rng = np.random.default_rng(10)
    def docs(n, lens):
        return [rng.integers(0, VOCAB, size=int(lens[i % len(lens)])).tolist() for i in range(n)]
    corpus = {"train": docs(12, [40, 0, 7, 95, 16, 3, 0, 33]), "test": docs(9, [25, 4, 0, 50, 2, 18])}
    as_text = lambda d: " ".join(f"w{t}" for t in d)
    out = {"vocab": VOCAB, "corpus": corpus, "cases": []}
    for ctx in (16, 32):
        task = TextTask.__new__(TextTask)
        task.context_length = ctx
        task.text_tokenizer = fast
        task.text_dataset = {k: datasets.Dataset.from_dict({"text": [as_text(d) for d in v]}) for k, v in corpus.items()}
        np.random.seed(70 + ctx)
        calls = []
        for bs, is_test in ((4, False), (9, False), (1, True), (6, True)):
            r = task.sample_batch(bs, is_test=is_test)
            calls.append({"batch_size": bs, "is_test": is_test, "out": [dict(d) for d in r]})
        np.random.seed(170 + ctx)
        ev = []
        for n in (5, 50):
            try:
                ev.append({"n": n, "metrics": task.evaluate(FakePolicy(), num_examples_to_test=n)})
            except ValueError:                   # np.random.randint(1, 1) on a one-token chunk: the reference raises
                ev.append({"n": n, "raises": "ValueError"})
        out["cases"].append({"context_length": ctx, "np_seed": 70 + ctx, "calls": calls, "eval_seed": 170 + ctx, "eval": ev})
    path = os.path.join(HERE, "g10_text_task.pt")
    torch.save(out, path)
    print(f"g10_text_task: {os.path.getsize(path) / 1024:.1f} KiB;", [[e.get('metrics', e.get('raises')) for e in c['eval']] for c in out['cases']])


if __name__ == "__main__":
    main()
