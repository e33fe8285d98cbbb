"""Twin of DRecPy's examples/consistency_eval/dmf.py: DMF with towers [128, 64] / [128, 64] (wider than a wavefront: a lane of the
fused step holds hidden units k and k + 64), with and without the normalised cross-entropy targets, leave-one-out by last timestamp,
HitRatio / NDCG @ 1..10 on one positive against 100 sampled negatives.
    python examples/consistency_eval_dmf.py [--movielens /data/ml-100k] [--epochs 50]"""
from _common import arguments, stopwatch

from drecpy_amd import synth
from drecpy_amd.Dataset import load_movielens
from drecpy_amd.Evaluation import leave_k_out, ranking_evaluation
from drecpy_amd.Evaluation.metrics import NDCG, HitRatio
from drecpy_amd.Recommender import DMF


def main():
    args = arguments(default_epochs=50, dataset_name='ml-100k')
    ds = load_movielens('ml-100k', args.movielens) if args.movielens else synth.dataset('ml-100k', extra_per_user=3)
    ds_train, ds_test = leave_k_out(ds, k=1, last_timestamps=True, seed=10)
    ds_train_bin = ds_train.copy()
    ds_train_bin.apply('interaction', lambda x: 1)
    ds_test_bin = ds_test.copy()
    ds_test_bin.apply('interaction', lambda x: 1)
    for nce in (True, False):
        print('NCE =', nce)
        dmf = DMF(use_nce=nce, user_factors=[128, 64], item_factors=[128, 64], seed=10, verbose=not args.quiet)
        with stopwatch(f'fit, {args.epochs} epochs of 256'):
            dmf.fit(ds_train if nce else ds_train_bin, epochs=args.epochs, batch_size=256, learning_rate=0.001, reg_rate=0.0001, neg_ratio=5)
        print(ranking_evaluation(dmf, ds_test if nce else ds_test_bin, n_pos_interactions=1, n_neg_interactions=100,
                                 generate_negative_pairs=True, novelty=True, k=list(range(1, 11)), metrics=[HitRatio(), NDCG()], seed=10,
                                 verbose=False))


if __name__ == '__main__':
    main()
