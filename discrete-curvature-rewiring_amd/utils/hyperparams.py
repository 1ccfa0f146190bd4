"""Per-dataset training and rewiring hyper-parameters — the values of the reference's utils/hyperparams.py:1-122
(tuned by Topping et al. / the thesis; they are data, reproduced as one row per dataset).

``hyperparams[name]`` is a dict with the reference's keys: dropout, hidden_depth, hidden_dim, learning_rate,
weight_decay (GCN / Adam, experiment/save_models.py:31-35,74-82) and max_iterations, tau, removal_bound (SDRF,
save_models.py:36-38,46).
"""
FIELDS = ('dropout', 'hidden_depth', 'hidden_dim', 'learning_rate', 'weight_decay', 'max_iterations', 'tau', 'removal_bound')

_ROWS = {
    'Cora'        : (0.3396, 1, 128, 0.0244, 0.1076, 100, 163, 0.95),
    'Citeseer'    : (0.4103, 1, 64, 0.0199, 0.4551, 84, 180, 0.22),
    'Pubmed'      : (0.3749, 3, 128, 0.0112, 0.0138, 166, 115, 14.43),
    'Cornell'     : (0.2911, 1, 128, 0.0056, 0.0366, 126, 145, 0.88),
    'Texas'       : (0.216, 1, 64, 0.0229, 0.0137, 89, 22, 1.64),
    'Wisconsin'   : (0.2452, 1, 64, 0.0113, 0.1559, 136, 12, 7.95),
    'Chameleon'   : (0.4886, 1, 32, 0.0268, 0.4056, 2441, 252, 2.84),
    'Squirrel'    : (0.3079, 1, 32, 0.0299, 0.0158, 1396, 436, 5.88),
    'Actor'       : (0.3424, 1, 64, 0.0129, 0.0126, 3249, 106, 0),
    'Computers'   : (0.3396, 1, 128, 0.0244, 0.1076, 100, 163, 0.95),
    'Photo'       : (0.3396, 1, 128, 0.0244, 0.1076, 100, 163, 0.95),
    'CoauthorCS'  : (0.3396, 1, 128, 0.0244, 0.1076, 100, 163, 0.95),
}

hyperparams = {name: dict(zip(FIELDS, row)) for name, row in _ROWS.items()}
