"""Generate tests/golden/g11_caption_vqa.pt by running the REFERENCE's CaptionTask / VqaTask sampling and evaluation
(gato/tasks/caption_task.py:112-159, gato/tasks/vqa_task.py:85-141).  Build container only.  Constructors (tar shards,
COCO folders, gpt2 download) are bypassed with ``__new__``; `dataset` and a real HF fast tokenizer are supplied; the
third-party imports the modules make at load time and this image lacks (webdataset, PIL) are empty stub modules.

    python tests/golden/make_fixture_caption_vqa.py
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
VOCAB = 40


class FakePolicy:
    device = "cpu"

    def __init__(self, tok):
        self.module, self.text_tokenizer = self, tok

    def _logits(self, image, prompt, max_length):
        g = torch.Generator().manual_seed(int(image.sum()) % 100003 + 7 * len(prompt) + max_length)
        return torch.randn(max_length, VOCAB, generator=g)

    def predict_caption(self, image, max_length=128, deterministic=True):
        return self._logits(image, [], max_length), "n/a"

    def predict_answer(self, image, question, max_length=16, deterministic=True):
        return self._logits(image, self.text_tokenizer.encode(question), max_length), "n/a"


def main():
    from make_fixture_sampler import install_stubs
    install_stubs()
    import importlib
    import types
    for name in ("PIL", "PIL.Image"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                m = types.ModuleType(name); m.Image = object; sys.modules[name] = m
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    from gato.tasks.caption_task import CaptionTask
    from gato.tasks.vqa_task import VqaTask

    vocab = {f"w{i}": i for i in range(VOCAB)}
    vocab["[UNK]"] = VOCAB
    tok = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="[UNK]")
    as_text = lambda ids: " ".join(f"w{t}" for t in ids)
    rng = np.random.default_rng(11)
    img = lambda: torch.tensor(rng.integers(0, 256, (1, 3, 16, 16)).astype(np.uint8))
    ids = lambda n: rng.integers(0, VOCAB, n).tolist()
    cap = {p: [{"image": img(), "text": ids(int(rng.integers(2, 9)))} for _ in range(n)] for p, n in (("train", 7), ("test", 4))}
    vqa = {p: [{"image": img(), "question": ids(int(rng.integers(2, 6))),
                "answers": [ids(int(rng.integers(1, 4))) for _ in range(int(rng.integers(1, 4)))]} for _ in range(n)]
           for p, n in (("train", 6), ("test", 5))}

    ct = CaptionTask.__new__(CaptionTask)
    ct.text_tokenizer = fast
    ct.dataset = {p: [{"image": e["image"], "text": as_text(e["text"])} for e in v] for p, v in cap.items()}
    vt = VqaTask.__new__(VqaTask)
    vt.text_tokenizer = fast
    vt.dataset = {p: [{"image": e["image"], "question": as_text(e["question"]),
                       "answers": [{"answer": as_text(a)} for a in e["answers"]]} for e in v] for p, v in vqa.items()}
    model = FakePolicy(fast)
    out = {"vocab": VOCAB, "caption": cap, "vqa": vqa, "calls": []}
    random.seed(2024)
    for bs in (3, 5):
        out["calls"].append(("caption_sample", bs, [{"images": d["images"].clone(), "text": list(d["text"])} for d in ct.sample_batch(bs)]))
        out["calls"].append(("vqa_sample", bs, [{"images": d["images"].clone(), "text": list(d["text"])} for d in vt.sample_batch(bs)]))
    out["calls"].append(("caption_eval", 3, ct.evaluate(model, num_examples_to_test=3)))
    out["calls"].append(("vqa_eval", 50, vt.evaluate(model, num_examples_to_test=50)))
    path = os.path.join(HERE, "g11_caption_vqa.pt")
    torch.save(out, path)
    print(f"g11_caption_vqa: {os.path.getsize(path) / 1024:.1f} KiB", [c[2] for c in out["calls"] if c[0].endswith("eval")])


if __name__ == "__main__":
    main()
