"""cProfile of NextFramePredictorS2S.train() as the notebook calls it (batch_size=1 loader): where the host time of a step goes.
    python tools/exp_train_profile.py [eager|graph]"""
import cProfile, io, os, pstats, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from torch.utils.data import DataLoader
from model.mpnnlstm import NextFramePredictorS2S
from helpers import TinyMovingMNISTDataset
dev = torch.device('cuda', 0)
torch.set_num_threads(min(16, os.cpu_count()))     # (a GPU box hands this job 16 cores of many: the DataLoader's collate otherwise spins ~13 ms per item)
ug = len(sys.argv) > 1 and sys.argv[1] == 'graph'
ds = TinyMovingMNISTDataset(64, 10, 10, n_digits=1, canvas_size=(64, 64), digit_size=(28, 28))
torch.manual_seed(0)
nfp = NextFramePredictorS2S(thresh=0.1, input_features=1, input_timesteps=10, output_timesteps=10, device=dev,
                            model_kwargs=dict(hidden_size=16, dropout=0.1, n_layers=2))
nfp.model.train()
train, test = DataLoader(ds, batch_size=1), DataLoader(torch.utils.data.Subset(ds, range(1)), batch_size=1)
so = sys.stdout; sys.stdout = open(os.devnull, 'w')
nfp.train(train, test, None, lr=0.0002, n_epochs=1, truncated_backprop=0, use_graph=ug)
pr = cProfile.Profile()
pr.enable()
nfp.train(train, test, None, lr=0.0002, n_epochs=2, truncated_backprop=0, use_graph=ug)
torch.cuda.synchronize()
pr.disable()
sys.stdout = so
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40)
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40)
print(s.getvalue())
