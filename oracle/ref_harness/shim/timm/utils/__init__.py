def get_state_dict(model, unwrap_fn=None):
    return model.state_dict()


class ModelEma:  # name only
    pass
