"""Experiment driver: CSV in -> fit -> predict -> .pred.csv / .loss.csv / .log.json out.

Behavioural counterpart of the reference's `scripts/evaluate.py:50-152` (run_experiment)
with its per-dataset tables (`scripts/dataset_config.py:19-180`, `scripts/evaluate.py:
194-302`), so that the published experiments run unchanged on the MI355X engine:

  python -m bayesnf_amd.evaluate --dataset chickenpox --objective map \
      --data_root DIR --output_dir OUT --start_id 8 --stop_id 9 [--num_particles 64]

Inputs  DIR/<dataset>.<series>.train.csv and .test.csv (index column 0, a `datetime`
column, the dataset's feature / target columns).  Outputs, per series, in OUT:
  bnf-<objective>.<dataset>.<series>.pred.csv   yhat (mean over members), yhat_p50,
                                                yhat_lower (2.5 %), yhat_upper (97.5 %)
  bnf-<objective>.<dataset>.<series>.loss.csv   one column per member, one row per epoch
  bnf-<objective>.<dataset>.<series>.log.json   runtime and the three config dicts
The tables below are data (hyper-parameters of the published runs), not code.
"""

from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np
import pandas as pd

from . import spatiotemporal

_COMMON = dict(timetype='index', feature_cols=['datetime', 'latitude', 'longitude'],
               standardize=['latitude', 'longitude'], num_series=10, series_id_fmt=str)
DATASET_CONFIG = {
    'air_quality': dict(_COMMON, target_col='pm10', freq='h'),
    'wind': dict(_COMMON, target_col='wind', freq='D'),
    'air': dict(_COMMON, target_col='pm10', freq='D'),
    'chickenpox': dict(_COMMON, target_col='chickenpox', freq='W'),
    'coprecip': dict(_COMMON, target_col='ppt', freq='M'),
    'sst': dict(_COMMON, target_col='sst', freq='M',
                feature_cols=['datetime', 'latitude', 'longitude', 'soi']),
}

# width, depth, seasonality periods, harmonics (same for map / mle / vi)
MODEL_CONFIG = {
    'air_quality': dict(width=512, depth=2, seasonality_periods=[24, 24 * 7],
                        num_seasonal_harmonics=[4, 4]),
    'wind': dict(width=512, depth=2, seasonality_periods=[7, 365.25 / 12, 365.25],
                 num_seasonal_harmonics=[3, 10, 10]),
    'air': dict(width=512, depth=2, seasonality_periods=[7, 365.25 / 12, 365.25],
                num_seasonal_harmonics=[3, 10, 10]),
    'chickenpox': dict(width=256, depth=2, seasonality_periods=[4.0, 52.1775],
                       num_seasonal_harmonics=[2.0, 10]),
    'coprecip': dict(width=512, depth=2, seasonality_periods=[12], num_seasonal_harmonics=[6]),
    'sst': dict(width=768, depth=2, seasonality_periods=[12], num_seasonal_harmonics=[6]),
}
MODEL_CONFIG = {ds: {obj: dict(cfg, observation_model='NORMAL') for obj in ('map', 'mle', 'vi')}
                for ds, cfg in MODEL_CONFIG.items()}

_MAP = lambda p, e, **kw: dict(num_particles=p, num_epochs=e, learning_rate=0.005, **kw)
_VI = lambda p, e, b, k, lr=0.01: dict(num_particles=p, num_epochs=e, learning_rate=lr, batch_size=b,
                                      kl_weight=k, sample_size_divergence=5)
INFERENCE_CONFIG = {
    'air_quality': {'map': _MAP(16, 4000, batch_size=38096), 'vi': _VI(16, 500, 3500, 0.2)},
    'wind': {'map': _MAP(64, 10000), 'vi': _VI(64, 2000, 3944, 0.1)},
    'air': {'map': _MAP(8, 7500), 'vi': _VI(8, 1000, 3800, 0.2)},
    'chickenpox': {'map': _MAP(64, 10000), 'vi': _VI(64, 1000, 511, 0.1)},
    'coprecip': {'map': _MAP(16, 7500), 'vi': _VI(16, 750, 3300, 0.2)},
    'sst': {'map': _MAP(16, 5000, batch_size=221127), 'vi': _VI(16, 600, 8845, 0.5, lr=0.005)},
}
for _cfg in INFERENCE_CONFIG.values():
  _cfg['mle'] = _cfg['map']

_ESTIMATORS = {'map': spatiotemporal.BayesianNeuralFieldMAP,
               'mle': spatiotemporal.BayesianNeuralFieldMLE,
               'vi': spatiotemporal.BayesianNeuralFieldVI}


def run_experiment(dataset, data_root, series_id, output_dir, objective, dataset_config,
                   model_config, inference_config, seed, compute_dtype=None):
  """One series: returns (losses, means, quantiles) and writes the three files."""
  if objective not in _ESTIMATORS:
    raise ValueError(f'objective={objective}')
  read = lambda split: pd.read_csv(
      os.path.join(data_root, f'{dataset}.{series_id}.{split}.csv'), index_col=0,
      parse_dates=['datetime'])
  df_train, df_test = read('train'), read('test')
  os.makedirs(output_dir, exist_ok=True)
  stem = os.path.join(output_dir, f'bnf-{objective}.{dataset}.{series_id}')
  model_kwargs = dict(model_config)
  model_kwargs.setdefault('observation_model', 'NORMAL')
  model_kwargs.update(feature_cols=dataset_config['feature_cols'],
                      target_col=dataset_config['target_col'],
                      timetype=dataset_config['timetype'], freq=dataset_config.get('freq'),
                      standardize=dataset_config.get('standardize'))
  fit_kwargs = dict(learning_rate=inference_config['learning_rate'],
                    num_epochs=inference_config['num_epochs'],
                    batch_size=inference_config.get('batch_size'),
                    ensemble_size=inference_config['num_particles'])
  if objective == 'vi':
    fit_kwargs.update(kl_weight=inference_config.get('kl_weight', 1.0),
                      sample_size_divergence=inference_config.get('sample_size_divergence', 10))
  else:
    fit_kwargs.update(num_splits=inference_config.get('num_particle_splits', 1))

  t0 = time.perf_counter()
  model = _ESTIMATORS[objective](compute_dtype=compute_dtype, **model_kwargs)
  model.fit(df_train, seed, **fit_kwargs)
  both = pd.concat([df_train, df_test])
  means, quantiles = model.predict(both, quantiles=(0.5, 0.025, 0.975))
  losses = model.losses_
  runtime = time.perf_counter() - t0

  with open(stem + '.log.json', 'w') as f:
    json.dump(dict(dataset=dataset, series_id=series_id, runtime=runtime, objective=objective,
                   dataset_config=dataset_config, model_config=model_kwargs,
                   inference_config=inference_config), f, indent=2, default=repr)
  pd.DataFrame(losses.reshape((-1, losses.shape[-1])).T).to_csv(stem + '.loss.csv', index=False)
  index = model.data_handler.copy_and_filter_table(both).index
  pred = pd.DataFrame({'yhat': np.mean(means, axis=tuple(range(means.ndim - 1))),
                       'yhat_p50': quantiles[0], 'yhat_lower': quantiles[1],
                       'yhat_upper': quantiles[2]}, index=index)
  pred.sort_index(inplace=True)
  pred.to_csv(stem + '.pred.csv', index=True)
  return losses, means, np.asarray(quantiles)


def main(argv=None):
  ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
  ap.add_argument('--output_dir', required=True)
  ap.add_argument('--data_root', required=True)
  ap.add_argument('--dataset', required=True, choices=sorted(DATASET_CONFIG))
  ap.add_argument('--objective', required=True, choices=sorted(_ESTIMATORS))
  ap.add_argument('--start_id', type=int, default=None)
  ap.add_argument('--stop_id', type=int, default=None)
  ap.add_argument('--num_particles', type=int, default=None)
  ap.add_argument('--seed', type=int, default=0)
  ap.add_argument('--compute_dtype', default=None, choices=[None, 'fp32', 'fp32_split', 'bf16', 'fp8'])
  args = ap.parse_args(argv)
  from . import distributed
  distributed.maybe_init_from_env()
  dcfg = DATASET_CONFIG[args.dataset]
  icfg = dict(INFERENCE_CONFIG[args.dataset][args.objective])
  if args.num_particles is not None:
    icfg['num_particles'] = args.num_particles
  start = 0 if args.start_id is None else args.start_id
  stop = dcfg['num_series'] if args.stop_id is None else args.stop_id
  for sid in range(start, stop):
    print(f'Running experiment {args.dataset}.{sid}', flush=True)
    run_experiment(args.dataset, args.data_root, dcfg['series_id_fmt'](sid), args.output_dir, args.objective,
                   dcfg, MODEL_CONFIG[args.dataset][args.objective], icfg, args.seed, compute_dtype=args.compute_dtype)


if __name__ == '__main__':
  main()
