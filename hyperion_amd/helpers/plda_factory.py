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
