"""CDAE in reference mode — the step DRecPy itself takes (every output unit, dense Keras Adam) — then HR/NDCG@k.
    python examples/cdae.py [--movielens /data/ml-100k] [--epochs 500]"""
from _common import arguments, split, stopwatch

from drecpy_amd.Evaluation import ranking_evaluation
from drecpy_amd.Recommender import CDAE


def main():
    args = arguments(default_epochs=500, dataset_name='ml-100k')
    train, test = split(args, 'ml-100k')
    model = CDAE(hidden_factors=50, corruption_level=0.2, loss='bce', seed=10, verbose=not args.quiet)
    with stopwatch(f'fit, {args.epochs} one-batch epochs of 64'):
        model.fit(train, epochs=args.epochs, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
    protocol = dict(k=[1, 5, 10], novelty=True, n_test_users=100, n_pos_interactions=1, n_neg_interactions=100,
                    generate_negative_pairs=True, seed=10, verbose=False)
    for name, value in ranking_evaluation(model, test, **protocol).items():
        print(f'  {name:14s} {value}')
    print('top-5 for raw user 1:', model.recommend(1, n=5))


if __name__ == '__main__':
    main()
