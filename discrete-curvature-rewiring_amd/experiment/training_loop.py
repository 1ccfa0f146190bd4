"""Full-batch training driver — call surface of the reference's experiment/training_loop.py:10-75.

``training_loop(model, optimizer, data, epochs, patience)``, ``train(model, optimizer, data)`` and
``evaluate(model, data, test)`` behave as the reference's do:

  * one epoch = one optimisation step on the training mask, then a validation forward pass;
  * the checkpoint moves forward whenever validation accuracy is >= the best so far (ties included);
  * after ``patience`` epochs without that, or after the last epoch, the best weights are restored;
  * ``evaluate`` reports ``val_acc`` (and ``test_acc`` when ``test`` is true) from arg-max predictions.

Any ``torch.nn.Module`` mapping ``data`` to log-probabilities works; the masks are read with
``data[f'{split}_mask']`` item access exactly as the reference does.
"""
import copy

import torch
import torch.nn.functional as F


def train(model, optimizer, data):
    """One optimisation step on the training nodes; returns the loss value (training_loop.py:40-54)."""
    model.train()
    optimizer.zero_grad()
    log_probs = model(data)
    mask = data.train_mask
    loss = F.nll_loss(log_probs[mask], data.y[mask])
    loss.backward()
    optimizer.step()
    return loss.item()


def _accuracy(log_probs, labels, mask):
    predicted = log_probs[mask].max(1)[1]
    return predicted.eq(labels[mask]).sum().item() / mask.sum().item()


def evaluate(model, data, test):
    """Accuracy on the validation split, plus the test split when ``test`` is true (training_loop.py:57-75)."""
    model.eval()
    with torch.no_grad():
        log_probs = model(data)
    splits = ('val', 'test') if test else ('val',)
    return {f'{split}_acc': _accuracy(log_probs, data.y, data[f'{split}_mask']) for split in splits}


def training_loop(model, optimizer, data, epochs, patience):
    """Train with early stopping on validation accuracy; returns the model holding the best weights
    (training_loop.py:10-37)."""
    best_acc, best_weights, since_best = 0, None, 0
    for _ in range(epochs):
        train(model, optimizer, data)
        val_acc = evaluate(model, data, test=False)['val_acc']
        if val_acc >= best_acc:  # ties advance the checkpoint, as in the reference
            best_acc, since_best = val_acc, 0
            best_weights = copy.deepcopy(model.state_dict())
        else:
            since_best += 1
        if since_best >= patience:
            break
    model.load_state_dict(best_weights)
    return model
