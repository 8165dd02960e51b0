"""`python -m trainer.logistic_matrix_factorisation` — the reference's second estimator on the same model
(reference src/models/logistic_matrix_factorisation.py:14-93).

Same MatrixFactorisation layer, regulariser, optimizers, flags and job_dir as `trainer.estimator`; the loss is
MultiHead([BinaryClassHead(weight_column=pos_name), BinaryClassHead(weight_column=neg_name)], [1, neg_factor])
on ONE logit with labels 1 / 0 (logistic_matrix_factorisation.py:48-54), i.e. per pair
`pos * softplus(-p) + neg_factor * neg * softplus(p)`, summed and divided by the batch size.  The input columns
are `[row_name, col_name, pos_name, neg_name]` with no label column (logistic_matrix_factorisation.py:66-70;
defaults `value` / `neg_weight`, configs/app.ini:35-36).  On the GPU it is an epilogue of the same pass kernel
(`glove_hyper.head = GLOVE_HEAD_LOGISTIC`, include/glove_hip.h).
"""
from trainer import estimator
from trainer.config_utils import save_params


def use_logistic_heads(params: dict) -> None:
    params["input_fn_args"].update({
        "select_columns": [params["row_name"], params["col_name"], params["pos_name"], params["neg_name"]],
        "target_names": [],
    })
    params["head"] = "logistic"
    save_params(params)


def main(argv=None):
    estimator.main(argv, adapt_params=use_logistic_heads)


if __name__ == "__main__":
    try:
        main()
    except KeyboardInterrupt:
        pass
