"""First-order architecture step (reference models/search/darts/architect.py:9-29)."""


class Architect(object):
    def __init__(self, model, args, criterion, optimizer):
        self.network_weight_decay = args.weight_decay
        self.criterion = criterion
        self.model = model
        self.optimizer = optimizer
        # opt-in (args.hip_graph): forward + criterion + backward + Adam(alpha) as one hipGraph replay
        self.use_graph = bool(getattr(args, 'hip_graph', False))
        self._graph = None
        self.graph_replays = 0

    def log_learning_rate(self, logger):
        for group in self.optimizer.param_groups:
            logger.info("Architecture Learning Rate: {}".format(group['lr']))
            break

    def step(self, input_valid, target_valid, logger):
        if self.use_graph:
            if self._graph is None:
                from bmnas.graph import GraphedTrainStep
                self._graph = GraphedTrainStep.try_build(self.model, self.criterion, self.optimizer,
                                                         input_valid, target_valid, logger)
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
