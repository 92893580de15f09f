class Policy:
    """Interface of all policies (/root/reference/go_ai/policies/__init__.py:1-20)."""

    def __init__(self, name, temp=None):
        self.name = name
        self.temp = temp
        self.pt_model = None

    def __call__(self, go_env, **kwargs):
        """go_env -> action probabilities [A]"""
        pass

    def __str__(self):
        return "{} {}".format(self.__class__.__name__, self.name)
