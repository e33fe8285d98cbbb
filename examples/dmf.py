"""DMF: two dense towers over a user's rating row and an item's rating column, cosine score (DRecPy/Recommender/dmf.py) — the
base class of the ModifiedDMF that DRecPy's examples/extending_recommender_dmf.py builds; its twin here is
examples/extending_recommender_dmf.py.  score_matrix() scores users against all items with the bf16 MFMA kernel.
    python examples/dmf.py [--movielens /data/ml-1m] [--epochs 200]"""
from _common import arguments, split, stopwatch

from drecpy_amd.Evaluation import ranking_evaluation
from drecpy_amd.Recommender import DMF


def main():
    args = arguments(default_epochs=200, dataset_name='ml-1m')
    train, test = split(args, 'ml-1m')
    model = DMF(user_factors=[64, 32], item_factors=[64, 32], seed=10, verbose=not args.quiet)
    with stopwatch(f'fit, {args.epochs} epochs of 256'):
        model.fit(train, epochs=args.epochs, batch_size=256, learning_rate=1e-3, reg_rate=1e-4, neg_ratio=5)
    scores = ranking_evaluation(model, test, k=[1, 5, 10], novelty=True, n_test_users=100, n_pos_interactions=1,
                                n_neg_interactions=100, generate_negative_pairs=True, seed=10, verbose=False)
    for name, value in scores.items():
        print(f'  {name:14s} {value}')


if __name__ == '__main__':
    main()
