import os, sys, cProfile, pstats, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.data_loader import Dataloader_hier_model_xing
from hiertcn_amd.data_prepare import load_xing
from hiertcn_amd.model import HierTCNModel
from hiertcn_amd.run_hier_xing import evaluate_hier
tmp = tempfile.mkdtemp()
subprocess.check_call([sys.executable, "scripts/make_synth_xing.py", "30000", tmp], stdout=subprocess.DEVNULL)
data = load_xing(os.path.join(tmp, "dataset", "interactions.csv"), max_activity=500)
n_items = int(data[1][:, 1].max()) + 1
d = 256
args = hier_tcn_args(item_num=n_items, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, batch_size=32, max_session_num=10, max_activity_len=20, load_worker_num=0)
args.log_path = None
model = HierTCNModel(args)
def run():
    ld = Dataloader_hier_model_xing(args, "all", data=data)
    return evaluate_hier(args, ld, np.zeros((32, 512), np.float32), model, prefix="test")
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
