"""First-order architecture step (reference models/search/darts/architect.py:9-29)."""


class Architect(object):
    def __init__(self, model, args, criterion, optimizer):
        self.network_weight_decay = args.weight_decay
        self.criterion = criterion
        self.model = model
        self.optimizer = optimizer

    def log_learning_rate(self, logger):
        for group in self.optimizer.param_groups:
            logger.info("Architecture Learning Rate: {}".format(group['lr']))
            break

    def step(self, input_valid, target_valid, logger):
        self.optimizer.zero_grad()
        self._backward_step(input_valid, target_valid)
        self.optimizer.step()

    def _backward_step(self, input_valid, target_valid):
        loss = self.criterion(self.model(input_valid), target_valid)
        loss.backward()
