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
        self._with_metric = False

    def log_learning_rate(self, logger):
        for group in self.optimizer.param_groups:
            logger.info("Architecture Learning Rate: {}".format(group['lr']))
            break

    def step(self, input_valid, target_valid, logger, metric=False):
        """metric=True (the trainer loop's dev phase): the caller will evaluate `criterion(model(input_valid),
        target_valid)` without gradients right after this step (train_searchable/mmimdb.py:66-84) — a captured step
        then carries that forward at its end and this returns (loss, output) of it; None: evaluate it yourself."""
        if self.use_graph:
            if self._graph is None and self._attempts < 3:
                from bmnas.graph import GraphedTrainStep
                self._attempts += 1
                self._with_metric = bool(metric)
                g = False
                if metric:
                    g = GraphedTrainStep.try_build(self.model, self.criterion, self.optimizer, input_valid,
                                                   target_valid, logger, metric_forward=True)
                    self._with_metric = bool(g)
                if not g:
                    g = GraphedTrainStep.try_build(self.model, self.criterion, self.optimizer, input_valid,
                                                   target_valid, logger)
                self._graph = g or None
            if self._graph and self._graph.matches(input_valid, target_valid):
                out = self._graph(input_valid, target_valid)
                self.graph_replays += 1
                if self._with_metric:
                    # (the replay ran the metric forward whether or not this caller wants it: BatchNorm statistics and
                    # the dropout stream advanced as they do when the caller runs it, which it then must not do again)
                    return out[2], out[3]
                return None
        self.optimizer.zero_grad()
        self._backward_step(input_valid, target_valid)
        self.optimizer.step()
        return None

    def step_k(self, batches, logger):
        """k architecture steps, each followed by its metric forward (`step(..., metric=True)` k times), as ONE hipGraph
        replay over k resident batches — the dev phase of a search with the trainer's `steps_per_replay` = k.
        -> [(loss, output)] of the k metric forwards, or None: the caller takes the batches one by one."""
        if not self.use_graph:
            return None
        k = len(batches)
        from bmnas.graph import GraphedTrainStep
        gk = getattr(self, '_graph_k', None)
        if gk is None and getattr(self, '_attempts_k', 0) < 3:
            self._attempts_k = getattr(self, '_attempts_k', 0) + 1
            gk = self._graph_k = GraphedTrainStep.try_build(self.model, self.criterion, self.optimizer, batches[0][0],
                                                            batches[0][1], logger, metric_forward=True, k=k) or None
        if not gk or gk.k != k or not all(gk.matches(x, y) for x, y in batches):
            return None
        for j, (x, y) in enumerate(batches):
            gk.stage(j, x, y)
        self.graph_replays += k
        return [(o[2], o[3]) for o in gk.replay_staged()]

    def _backward_step(self, input_valid, target_valid):
        loss = self.criterion(self.model(input_valid), target_valid)
        loss.backward()
