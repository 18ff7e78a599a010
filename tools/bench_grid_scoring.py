"""Cost of the reference's five EpochScoring metrics (config-transformer.yaml:9) in the grid: folds/hr with and without."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
import torch
from slnlp.data import synthetic_dataset
from slnlp.grid import ShardedGridSearchCV
from slnlp.net import NeuralNetClassifier
warnings.filterwarnings("ignore")
SCORING = ["neg_log_loss", "accuracy", "precision_weighted", "recall_weighted", "f1_weighted"]
ds = synthetic_dataset(2000, seq_len=48, src_vocab=3000, n_labels=200, seed=1, min_len=8)
for scoring in (None, SCORING):
    for k in (1, 2, 3, 4, 6):
        factory = lambda: NeuralNetClassifier(
            module="model.Transformer", module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y, module__batch_first=True,
            module__embedding_size=128, module__num_heads=4, module__num_layers=2, module__hidden_size=256, module__dropout=0.1,
            criterion__ignore_index=1, optimizer__momentum=0.9, optimizer__nesterov=False, lr=0.01, max_epochs=10, batch_size=50,
            device="cuda:0", gradient_clipping={"gradient_clip_value": 0.5}, scoring=scoring, train_split=5)
        grid = {"lr": [0.1, 0.01, 0.001], "module__embedding_size": [128, 512]}
        t0 = time.perf_counter()
        gs = ShardedGridSearchCV(factory, grid, cv=2, refit=False, device="cuda:0", fits_per_gpu=k).fit(ds)
        dt = time.perf_counter() - t0
        print(f"scoring={'5 metrics' if scoring else 'none':9s} fits_per_gpu={k}: {gs.n_tasks_ / dt * 3600:8.0f} folds/hr  ({dt:.2f} s for {gs.n_tasks_} fits)", flush=True)
