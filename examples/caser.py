"""Caser: horizontal / vertical convolutions over a user's last L items, trained from ListSampler windows ordered by
timestamp; scored with recommendation_evaluation (top-N against the held-out items).
    python examples/caser.py [--movielens /data/ml-1m] [--epochs 100]"""
from _common import arguments, split, stopwatch

from drecpy_amd.Evaluation import recommendation_evaluation
from drecpy_amd.Recommender import Caser


def main():
    args = arguments(default_epochs=100, dataset_name='ml-1m')
    train, test = split(args, 'ml-1m', min_user_interactions=20)
    model = Caser(L=5, T=3, d=50, n_v=4, n_h=16, dropout_rate=0.5, sort_column='timestamp', seed=10, verbose=not args.quiet)
    with stopwatch(f'fit, {args.epochs} epochs of 512 windows'):
        model.fit(train, epochs=args.epochs, batch_size=512, learning_rate=1e-3, reg_rate=1e-6, neg_ratio=3)
    scores = recommendation_evaluation(model, test, n_test_users=200, k=[1, 5, 10], novelty=True, seed=10, verbose=False)
    for name, value in scores.items():
        print(f'  {name:14s} {value}')


if __name__ == '__main__':
    main()
