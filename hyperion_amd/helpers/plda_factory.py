from ..pdfs.plda import FRPLDA, PLDA, SPLDA


class PLDAFactory(object):
    """Model construction by type name (hyperion/helpers/plda_factory.py:14-71)."""

    _classes = {"frplda": FRPLDA, "splda": SPLDA, "plda": PLDA}

    @staticmethod
    def create_plda(plda_type, y_dim=None, z_dim=None, fullcov_W=True, update_mu=True, update_V=True,
                    update_U=True, update_B=True, update_W=True, update_D=True, floor_iD=1e-5, name="plda",
                    **kwargs):
        if plda_type == "frplda":
            return FRPLDA(fullcov_W=fullcov_W, update_mu=update_mu, update_B=update_B, update_W=update_W,
                          name=name, **kwargs)
        if plda_type == "splda":
            return SPLDA(y_dim=y_dim, fullcov_W=fullcov_W, update_mu=update_mu, update_V=update_V,
                         update_W=update_W, name=name, **kwargs)
        if plda_type == "plda":
            return PLDA(y_dim=y_dim, z_dim=z_dim, floor_iD=floor_iD, update_mu=update_mu, update_V=update_V,
                        update_U=update_U, update_D=update_D, name=name, **kwargs)
        raise ValueError("unknown PLDA type %r" % (plda_type,))

    @staticmethod
    def load_plda(plda_type, model_file):
        if plda_type not in PLDAFactory._classes:
            raise ValueError("unknown PLDA type %r" % (plda_type,))
        return PLDAFactory._classes[plda_type].load(model_file)

    # ---- command-line plumbing of the training and evaluation scripts ----------------------------
    _train_args = ("plda_type", "y_dim", "z_dim", "diag_W", "no_update_mu", "no_update_V", "no_update_U",
                   "no_update_B", "no_update_W", "no_update_D", "floor_iD", "epochs", "ml_md", "md_epochs", "name")
    _negated = (("diag_W", "fullcov_W"), ("no_update_mu", "update_mu"), ("no_update_V", "update_V"),
                ("no_update_U", "update_U"), ("no_update_B", "update_B"), ("no_update_W", "update_W"),
                ("no_update_D", "update_D"))

    @staticmethod
    def filter_train_args(prefix=None, **kwargs):
        """The create_plda / fit arguments among a parsed command line, the negated switches turned
        into their positive names (hyperion/helpers/plda_factory.py:73-115; the loop there names an
        undefined variable and stops with a NameError - its evident intent is what runs here, and a
        switch that is absent from kwargs is simply left out)."""
        d = dict((k, kwargs[k]) for k in PLDAFactory._train_args if k in kwargs)
        for neg, pos in PLDAFactory._negated:
            if neg in d:
                d[pos] = not d.pop(neg)
        return d

    @staticmethod
    def add_class_args(parser, prefix=None):
        """Training options (plda_factory.py:118-191), defaults and spellings as there - including
        `--diag-W`, a store_false switch whose default is already False, and `--md-epochs`, which has
        no prefix."""
        p1 = "--" if prefix is None else "--" + prefix + "."
        parser.add_argument(p1 + "plda-type", default="splda", choices=["frplda", "splda", "plda"], help="PLDA type")
        parser.add_argument(p1 + "y-dim", type=int, default=150, help="num. of eigenvoices")
        parser.add_argument(p1 + "z-dim", type=int, default=400, help="num. of eigenchannels")
        parser.add_argument(p1 + "diag-W", default=False, action="store_false", help="use diagonal covariance W")
        for what in ("mu", "V", "U", "B", "W", "D"):
            parser.add_argument(p1 + "no-update-" + what, default=False, action="store_true",
                                help="not update " + what)
        parser.add_argument(p1 + "floor-iD", type=float, default=1e-5, help="floor for inverse of D matrix")
        parser.add_argument(p1 + "epochs", type=int, default=40, help="num. of epochs")
        parser.add_argument(p1 + "ml-md", default="ml+md", choices=["ml+md", "ml", "md"], help="optimization type")
        parser.add_argument("--md-epochs", default=None, type=int, nargs="+",
                            help="epochs in which we do MD, if None we do it in all the epochs")
        parser.add_argument(p1 + "name", default="plda", help="model name")

    @staticmethod
    def filter_eval_args(prefix=None, **kwargs):
        """plda_factory.py:193-196."""
        return dict((k, kwargs[k]) for k in ("plda_type", "model_file") if k in kwargs)

    @staticmethod
    def add_eval_args(parser, prefix=None):
        """plda_factory.py:198-211."""
        p1 = "--" if prefix is None else "--" + prefix + "."
        parser.add_argument(p1 + "plda-type", default="splda", choices=["frplda", "splda", "plda"], help="PLDA type")
        parser.add_argument(p1 + "model-file", required=True, help="model file")

    add_argparse_train_args = add_class_args
    add_argparse_eval_args = add_eval_args
