"""cProfile of ONE small fit (E128 N2, 10 epochs, 1000 samples): where does the host spend its time?"""
import cProfile, os, pstats, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
import torch
from slnlp.data import synthetic_dataset
from slnlp.net import NeuralNetClassifier
warnings.filterwarnings("ignore")
ds = synthetic_dataset(1000, seq_len=48, src_vocab=3000, n_labels=200, seed=1, min_len=8)
def make(E=128):
    return NeuralNetClassifier(
        module="model.Transformer", module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y, module__batch_first=True,
        module__embedding_size=E, module__num_heads=4, module__num_layers=2, module__hidden_size=256, module__dropout=0.1,
        criterion__ignore_index=1, optimizer__momentum=0.9, optimizer__nesterov=False, lr=0.01, max_epochs=10, batch_size=50,
        device="cuda:0", gradient_clipping={"gradient_clip_value": 0.5},
        scoring=["neg_log_loss", "accuracy", "precision_weighted", "recall_weighted", "f1_weighted"])
make().fit(ds)                      # warm: library load, first-touch allocations
for E in (128, 512):
    t0 = time.perf_counter(); n = make(E); n.initialize(); t1 = time.perf_counter(); n.fit(ds); t2 = time.perf_counter()
    print(f"E{E}: initialize {1e3 * (t1 - t0):.1f} ms, fit {1e3 * (t2 - t1):.1f} ms ({sum(r['dur'] for r in n.history) * 1e3:.1f} ms in epochs)")
pr = cProfile.Profile(); pr.enable(); make().fit(ds); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
