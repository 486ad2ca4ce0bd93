"""Learning-rate schedules and optimizer descriptors with the Keras names tools/train.py uses
(reference tools/train.py:31-53).  They only carry hyper-parameters; the update itself is the fused
HIP kernel (ssd_adam_step / ssd_sgd_step) driven by SSDEngine."""


class ExponentialDecay:
    """tf.keras.optimizers.schedules.ExponentialDecay, staircase=False: lr0 * rate ** (step / decay_steps)."""

    def __init__(self, initial_learning_rate, decay_steps, decay_rate):
        self.initial_learning_rate, self.decay_steps, self.decay_rate = initial_learning_rate, decay_steps, decay_rate

    def __call__(self, step):
        return self.initial_learning_rate * self.decay_rate ** (step / self.decay_steps)


class PolynomialDecay:
    """tf.keras.optimizers.schedules.PolynomialDecay, power=1, cycle=False."""

    def __init__(self, initial_learning_rate, decay_steps, end_learning_rate=0.0001, power=1.0):
        self.initial_learning_rate, self.decay_steps = initial_learning_rate, decay_steps
        self.end_learning_rate, self.power = end_learning_rate, power

    def __call__(self, step):
        s = min(step, self.decay_steps)
        return (self.initial_learning_rate - self.end_learning_rate) * (1 - s / self.decay_steps) ** self.power \
            + self.end_learning_rate


class _Optimizer:
    def __init__(self, learning_rate):
        self._lr = learning_rate
        self.iterations = 0                       # Keras: optimizer.iterations

    def lr(self, step=None):
        """Learning rate at the optimizer's own iteration count (Keras evaluates schedules there)."""
        it = self.iterations if step is None else step
        return self._lr(it) if callable(self._lr) else float(self._lr)


class Adam(_Optimizer):
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, name="Adam", **_):
        super().__init__(learning_rate)
        self.beta_1, self.beta_2, self.epsilon, self.name = beta_1, beta_2, epsilon, name


class SGD(_Optimizer):
    def __init__(self, learning_rate=0.01, name="SGD", **_):
        super().__init__(learning_rate)
        self.name = name
