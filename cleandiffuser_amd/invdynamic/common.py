class BasicInvDynamic:
    """Protocol of an inverse-dynamics head: ``predict(...)`` and call-through (reference invdynamic/common.py:1-6)."""

    def predict(self, **kwargs):
        raise NotImplementedError

    def __call__(self, **kwargs):
        return self.predict(**kwargs)
