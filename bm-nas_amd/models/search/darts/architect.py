"""First-order architecture step (reference models/search/darts/architect.py:9-29)."""


class Architect(object):
    def __init__(self, model, args, criterion, optimizer):
        self.network_weight_decay = args.weight_decay
        self.criterion = criterion
        self.model = model
        self.optimizer = optimizer
        # forward + criterion + backward + Adam(alpha) as one hipGraph replay (GraphedTrainStep.enabled)
        from bmnas.graph import GraphedTrainStep
        self.use_graph = GraphedTrainStep.enabled(args)
        self._graph = None
        self._attempts = 0
        self.graph_replays = 0

    def log_learning_rate(self, logger):
        for group in self.optimizer.param_groups:
            logger.info("Architecture Learning Rate: {}".format(group['lr']))
            break

    def step(self, input_valid, target_valid, logger):
        if self.use_graph:
            if self._graph is None and self._attempts < 3:
                from bmnas.graph import GraphedTrainStep
                self._attempts += 1
                self._graph = GraphedTrainStep.try_build(self.model, self.criterion, self.optimizer,
                                                         input_valid, target_valid, logger) or None
            if self._graph and self._graph.matches(input_valid, target_valid):
                self._graph(input_valid, target_valid)
                self.graph_replays += 1
                return
        self.optimizer.zero_grad()
        self._backward_step(input_valid, target_valid)
        self.optimizer.step()

    def _backward_step(self, input_valid, target_valid):
        loss = self.criterion(self.model(input_valid), target_valid)
        loss.backward()
