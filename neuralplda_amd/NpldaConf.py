"""Experiment configuration — counterpart of utils/NpldaConf.py:12-56 (same INI sections and keys, same
attribute names), selecting this package's score-file writers.  Differences: missing `cmiss`, `cfa`
and the subsample keys fall back to 1 / 1 / None (conf/sre18_egs_config.cfg lacks them and makes the
reference raise KeyError), and a readable error is raised for a missing file."""
import configparser as cp
import os

from .scorefile_generator import generate_sre_scores, generate_voices_scores

__all__ = ["NpldaConf"]


def _floats(s):
    return [float(x) for x in s.split(',')]


class NpldaConf:
    def __init__(self, configfile):
        if not os.path.exists(configfile):
            raise IOError(f"config file {configfile!r} not found")
        config = cp.ConfigParser(interpolation=cp.ExtendedInterpolation())
        config.read(configfile)
        P, N, T = config['Paths'], config['NPLDA'], config['Training']
        self.training_data_trials_list = P['training_data_trials_list'].split(',')
        self.validation_trials_list = P['validation_trials_list'].split(',')
        self.test_trials_list = P['test_trials_list'].split(',')
        self.mega_xvector_scp = P['mega_xvector_scp']
        self.mega_xvector_pkl = P['mega_xvector_pkl']
        self.meanvec = P['meanvec']
        self.transformmat = P['transformmat']
        self.kaldiplda = P['kaldiplda']
        self.xvector_dim = int(N['xvector_dim'])
        self.layer1_LDA_dim = int(N['layer1_LDA_dim'])
        self.layer2_PLDA_spkfactor_dim = int(N['layer2_PLDA_spkfactor_dim'])
        self.initialization = N['initialization']
        self.device = N['device']
        self.seed = int(N['seed'])
        self.alpha = float(N['alpha'])
        self.loss = T['loss']
        self.cmiss = float(T.get('cmiss', '1'))
        self.cfa = float(T.get('cfa', '1'))
        self.target_probs = T['target_probs'].split(',')
        # beta = cfa (1 - pt) / (cmiss pt) per target prior (utils/NpldaConf.py:38)
        self.beta = [self.cfa * (1 - float(pt)) / (self.cmiss * float(pt)) for pt in self.target_probs]
        self.batch_size = int(T['batch_size'])
        self.n_epochs = int(T['n_epochs'])
        self.lr = float(T['lr'])
        self.heldout_set_for_lr_decay = T['heldout_set_for_lr_decay']
        self.heldout_set_for_th_init = T['heldout_set_for_th_init']
        self.log_interval = int(config['Logging']['log_interval'])
        if config['Scoring']['scorefile_format'] == 'sre':
            self.generate_scorefile = generate_sre_scores
        else:
            self.generate_scorefile = generate_voices_scores
        tsf = T.get('train_subsample_factors', 'None')
        vsf = T.get('valid_subsample_factors', 'None')
        self.train_subsample_factors = None if tsf == 'None' else _floats(tsf)
        self.valid_subsample_factors = None if vsf == 'None' else _floats(vsf)
